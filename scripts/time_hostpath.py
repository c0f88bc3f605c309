# Where a host-to-host query (bn_bp_run_view) spends its time, run on the GPU box:
#   python scripts/time_hostpath.py [rows cols]
# Steps timed separately (each ends with its own synchronisation), then the fused call.
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (316, 316)
g = synth.grid(rows, cols, 4, seed=2)
evs = [synth.random_evidence(g, 0.01, seed=7 + q) for q in range(8)]
reps = 40


def avg(fn):
    for i in range(3):
        fn(i)
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    return (time.perf_counter() - t0) / reps * 1e6


out = {}
with Engine(g) as e:
    for flow in (1, 0):
        e.set_option("flow", flow)
        o = {}
        o["set_evidence_us"] = avg(lambda i: e.bp_set_evidence(evs[i % 8]))
        o["run_device_us"] = avg(lambda i: e.bp_run_device(1e-3))
        o["copy_beliefs_pageable_us"] = avg(lambda i: e.bp_beliefs())
        o["bp_run_pageable_us"] = avg(lambda i: e.bp_run(evs[i % 8], 1e-3))
        o["bp_run_view_us"] = avg(lambda i: e.bp_run_view(evs[i % 8], 1e-3))
        e.set_option("beliefs_direct", 1)
        o["bp_run_view_direct_us"] = avg(lambda i: e.bp_run_view(evs[i % 8], 1e-3))
        want = e.bp_run(evs[3], 1e-3)["beliefs"]
        import numpy as np
        assert np.array_equal(e.bp_run_view(evs[3], 1e-3)["beliefs"], want)
        e.set_option("beliefs_direct", 0)
        o["sweeps"] = e.bp_run_device(1e-3)["sweeps"]
        o["devclock_us_per_sweep"] = e.bp_stats()["sweep_devclock_ms"] * 1e3 / o["sweeps"]
        out[f"flow{flow}"] = o
print(json.dumps(out, indent=1))

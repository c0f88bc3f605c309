# Pause between two polls of a waiting tile (dataflow form of the resident kernel) vs. time per sweep; run on the GPU box.
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

for rows in (316, 200, 128):
    g = synth.grid(rows, rows, 4, seed=2)
    ev = synth.random_evidence(g, 0.01, seed=7)
    with Engine(g) as e:
        e.set_option("multisweep", 2)
        e.set_option("flow", 1)
        e.bp_set_evidence(ev)
        row = {}
        for z in (0, 1, 2, 4, 6, 8, 12, 16, 24):
            e.set_option("poll_sleep", z)
            for _ in range(3):
                e.bp_run_device(1e-3)
            t0 = time.perf_counter()
            dev, reps = 0.0, 30
            for _ in range(reps):
                r = e.bp_run_device(1e-3)
                dev += e.bp_stats()["sweep_devclock_ms"]
            wall = (time.perf_counter() - t0) / reps * 1e6
            assert e.info("last_flow") == 1
            row[z] = (round(dev / reps * 1e3 / r["sweeps"], 2), round(wall, 1))
        e.set_option("flow", 0)
        for _ in range(3):
            e.bp_run_device(1e-3)
        t0 = time.perf_counter()
        dev = 0.0
        for _ in range(30):
            r = e.bp_run_device(1e-3)
            dev += e.bp_stats()["sweep_devclock_ms"]
        wall = (time.perf_counter() - t0) / 30 * 1e6
        print(rows, "sweeps", r["sweeps"], "poll_sleep -> (us per sweep devclock, us per run wall):", json.dumps(row),
              "barrier:", (round(dev / 30 * 1e3 / r["sweeps"], 2), round(wall, 1)), flush=True)

#!/usr/bin/env python3
"""Networks of arities 2..4 on the register-resident DAG path ("dag" = 2: tables padded to four states) against the path the library
takes by default (item kernels / tiles): us per sweep (device clock), us per query.  GPU box only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402


def measure(eng, ev, eps, reps=40):
    eng.bp_set_evidence(ev)
    for _ in range(5):
        eng.bp_run_device(eps)
    t0 = time.perf_counter()
    dev, sweeps = 0.0, 0
    for _ in range(reps):
        r = eng.bp_run_device(eps)
        dev += eng.bp_stats()["sweep_devclock_ms"]
        sweeps += r["sweeps"]
    dt = time.perf_counter() - t0
    return {"path": eng.last_path(), "us_per_sweep": round(dev / sweeps * 1e3, 2), "us_per_query": round(dt / reps * 1e6, 1), "sweeps": sweeps / reps}


nets = [("mixed 2-4, <=3 parents, 300 nodes", synth.random_dag(300, 3, 32, [2, 3, 4], seed=5)),
        ("mixed 2-4, <=3 parents, 3 000 nodes", synth.random_dag(3000, 3, 64, [2, 3, 4], seed=6)),
        ("mixed 2-4, <=3 parents, 10 000 nodes", synth.random_dag(10000, 3, 64, [2, 3, 4], seed=7)),
        ("mixed 2-4, <=4 parents, 10 000 nodes", synth.random_dag(10000, 4, 64, [2, 3, 4], seed=8)),
        ("binary, <=4 parents, 10 000 nodes", synth.random_dag(10000, 4, 64, 2, seed=9)),
        ("k = 3 grid 64 x 64", synth.grid(64, 64, 3, seed=3)),
        ("k = 2 grid 128 x 128", synth.grid(128, 128, 2, seed=3))]
out = {}
for name, g in nets:
    ev = synth.random_evidence(g, 0.01, seed=7)
    with Engine(g) as eng:
        row = {"cpt_entries": int(g.cpt_off[-1]), "dag_blocks": eng.info("dag_blocks"), "dag_stream": eng.info("dag_stream")}
        row["default"] = measure(eng, ev, 1e-3)
        eng.set_option("dag", 2)
        row["dag"] = measure(eng, ev, 1e-3)
        eng.set_option("dag", 1)
        eng.set_option("autotune", 1)
        eng.bp_run_device(1e-3)
        row["autotuned_path"] = eng.info("autotuned_path")
    out[name] = row
    print(name, json.dumps(row), flush=True)
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)

#!/usr/bin/env python3
"""us per sweep (device clock) and us per query of the register-resident DAG path (bn_dag.hip) against the other paths on
k = 4 networks with up to 5 parents per node.  GPU box only:  python scripts/time_dag.py [--json out.json]"""
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
    return {"path": eng.last_path(), "us_per_sweep": dev / sweeps * 1e3, "us_per_query": dt / reps * 1e6, "sweeps": sweeps / reps}


def main():
    nets = [("dag10k (configs[1])", synth.random_dag(10000, 4, 64, 4, seed=1)),
            ("dag3000", synth.random_dag(3000, 4, 64, 4, seed=8)),
            ("dag1000", synth.random_dag(1000, 4, 64, 4, seed=9)),
            ("dag300", synth.random_dag(300, 4, 32, 4, seed=5)),
            ("dag200_p5", synth.random_dag(200, 5, 32, 4, seed=6)),
            ("grid64", synth.grid(64, 64, 4, seed=5)),
            ("dag15k", synth.random_dag(15000, 4, 64, 4, seed=3)),
            ("dag30k", synth.random_dag(30000, 4, 64, 4, seed=2))]
    if "--big" in sys.argv:
        nets.append(("dag100k", synth.random_dag(100000, 4, 64, 4, seed=11)))
    out = {}
    for name, g in nets:
        ev = synth.random_evidence(g, 0.01, seed=7)
        row = {}
        with Engine(g) as eng:
            row["info"] = {k: eng.info(k) for k in ("dag_eligible", "dag_blocks", "dag_tiles", "dag_stream", "mid_eligible", "resident_eligible")}
            for label, opts in (("dag", {"dag": 2}), ("default", {"dag": 1}), ("no_dag", {"dag": 0}), ("launches", {"dag": 0, "multisweep": 0})):
                for k, v in {"dag": 1, "multisweep": 1, **opts}.items():
                    eng.set_option(k, v)
                row[label] = measure(eng, ev, 1e-3)
            row["aborts"] = eng.info("dag_aborts")
        out[name] = row
        print(name, json.dumps(row), flush=True)
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()

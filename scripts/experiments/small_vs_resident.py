# Networks both the one-workgroup path (bn_small.hip) and the resident-tile kernel (bn_resident.hip) can run: which is faster?
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

nets = [("pearl", synth.pearl()), ("grid6k4", synth.grid(6, 6, 4, seed=1)), ("grid8k4", synth.grid(8, 8, 4, seed=1)), ("grid8k2", synth.grid(8, 8, 2, seed=1)),
        ("grid16k2", synth.grid(16, 16, 2, seed=1)), ("grid24k2", synth.grid(24, 24, 2, seed=1)), ("grid10k3", synth.grid(10, 10, 3, seed=1)), ("grid5k4", synth.grid(5, 5, 4, seed=1)),
        ("chain50k4", synth.grid(50, 1, 4, seed=5)), ("chain100k4", synth.grid(100, 1, 4, seed=5)), ("chain200k4", synth.grid(200, 1, 4, seed=5)),
        ("chain200k2", synth.grid(200, 1, 2, seed=5)), ("chain400k2", synth.grid(400, 1, 2, seed=5)),
        ("tree100k3", synth.random_dag(100, 1, 8, 3, seed=3)), ("tree200k4", synth.random_dag(200, 1, 8, 4, seed=3)), ("dag60k3p2", synth.random_dag(60, 2, 8, 3, seed=3)),
        ("dag100k2p2", synth.random_dag(100, 2, 8, 2, seed=3)), ("dag40k4p2", synth.random_dag(40, 2, 8, 4, seed=3))]
for name, mod in nets:
    with Engine(mod, lanes_per_node=2) as e:
        ev = synth.random_evidence(mod, 0.05, seed=3)
        e.bp_set_evidence(ev)
        res = {}
        for form in ("small", "resident"):
            e.set_option("multisweep", 2)
            e.set_option("small", 2 if form == "small" else 0)
            for _ in range(3):
                r = e.bp_run_device(1e-6)
            reps = 50
            t0 = time.perf_counter()
            dev = 0.0
            for _ in range(reps):
                r = e.bp_run_device(1e-6)
                dev += e.bp_stats()["sweep_devclock_ms"]
            res[form] = (e.last_path(), round(dev / reps * 1e3 / r["sweeps"], 2), round((time.perf_counter() - t0) / reps * 1e6, 1))
        p = e.small_plan()
        print(name, "nodes", mod.n, "entries", len(mod.cpt), "tiles", e.layout()["n_tiles"], "resident_blocks", e.info("resident_blocks"),
              "small", None if p is None else (p["waves"], p["re"], p["rb"], p["rc"]), res, flush=True)

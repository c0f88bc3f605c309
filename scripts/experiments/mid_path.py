# Mid-size networks: the tile kernels vs the item kernel spread over several workgroups (bn_mid.hip); run on the GPU box.
# (the oracle is the checker here, nothing of it is timed)
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402
import oracle  # noqa: E402

nets = [(f"mixed{n}", synth.random_dag(n, mp, 16, [2, 3, 4, 3, 2, 4, 5], seed=seed)) for n, mp, seed in ((80, 3, 10), (150, 3, 11), (300, 3, 12), (600, 3, 13), (1000, 3, 14))]
nets += [("dag60k4", synth.random_dag(60, 4, 16, 4, seed=5)), ("grid12k3", synth.grid(12, 12, 3, seed=1)), ("grid16k4", synth.grid(16, 16, 4, seed=1)),
         ("grid32k4", synth.grid(32, 32, 4, seed=1)), ("dag200k4p2", synth.random_dag(200, 2, 16, 4, seed=5)), ("dag1000k2p3", synth.random_dag(1000, 3, 16, 2, seed=5)),
         ("chain400k4", synth.grid(400, 1, 4, seed=1)), ("dag200k4", synth.random_dag(200, 4, 64, 4, seed=200)),
         ("mixed3000", synth.random_dag(3000, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=15)), ("mixed2000", synth.random_dag(2000, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=16)),
         # 96-128 parts at the largest part size
         ("mixed2k4p", synth.random_dag(2000, 4, 64, [2, 3, 4, 3, 2, 4, 4], seed=9)), ("dag800k4", synth.random_dag(800, 4, 48, 4, seed=77)),
         ("dag600k4", synth.random_dag(600, 4, 48, 4, seed=78)), ("mixed5000", synth.random_dag(5000, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=17)),
         ("k5p3_1200", synth.random_dag(1200, 3, 32, 5, seed=18)),
         # 200+ parts
         ("mixed8000", synth.random_dag(8000, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=19)), ("mixed10000", synth.random_dag(10000, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=20)),
         ("dag1000k4", synth.random_dag(1000, 4, 48, 4, seed=77)), ("dag800k4b", synth.random_dag(800, 4, 48, 4, seed=77)), ("dag400k4", synth.random_dag(400, 4, 48, 4, seed=79)), ("mixed4k4p", synth.random_dag(4000, 4, 64, [2, 3, 4, 3, 2, 4, 4], seed=21))]
if len(sys.argv) > 1:
    nets = [x for x in nets if x[0] in sys.argv[1:]]
for name, mod in nets:
    with Engine(mod) as e:
        ev = synth.random_evidence(mod, 0.05, seed=3)
        e.bp_set_evidence(ev)
        res = {}
        for form in ("tiles", "mid"):
            e.set_option("mid", 2 if form == "mid" else 0)   # 2: also where the policy would keep the tiles
            for _ in range(3):
                r = e.bp_run_device(1e-6)
            reps = 30
            t0 = time.perf_counter()
            dev = 0.0
            for _ in range(reps):
                r = e.bp_run_device(1e-6)
                dev += e.bp_stats()["sweep_devclock_ms"]
            wall = (time.perf_counter() - t0) / reps
            res[form] = {"path": e.last_path(), "sweeps": r["sweeps"], "us_sweep": round(dev / reps * 1e3 / r["sweeps"], 2),
                         "us_run_wall": round(wall * 1e6, 1), "beliefs": e.bp_beliefs(), "res": e.bp_residuals(), "msg": e.bp_messages()}
        o = oracle.bp_run(mod, ev, 1e-6, dump_msgs=True)
        m = res["mid"]
        exact = (np.array_equal(m["beliefs"], o["beliefs"], equal_nan=True) and m["sweeps"] == o["sweeps"] and np.array_equal(m["res"], o["residuals"])
                 and np.array_equal(m["msg"][0], o["pi_msg"]) and np.array_equal(m["msg"][1], o["lambda_msg"]))
        print(name, "nodes", mod.n, "entries", len(mod.cpt), "parts", e.info("mid_parts"), "aborts", e.info("mid_aborts"), "oracle_exact", exact,
              "max|d|", float(np.nanmax(np.abs(m["beliefs"] - o["beliefs"]))),
              {f: {k: res[f][k] for k in ("path", "sweeps", "us_sweep", "us_run_wall")} for f in res}, flush=True)

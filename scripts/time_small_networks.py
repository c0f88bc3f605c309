# Small networks: one launch per sweep vs. the whole run in one launch with resident tiles (bn_resident.hip); run on the GPU box.
# Prints, per network, sweeps, device-clock microseconds per sweep and host wall microseconds per run
# on both paths, and checks that the two paths give the same bits.
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bayesiannetwork_amd import Evidence, synth  # noqa: E402
from bayesiannetwork_amd.dsc import load_dsc  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

alarm, _ = load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))
nets = [("alarm_shaped", alarm), ("pearl", synth.pearl()), ("resume_chain", synth.resume_chain()),
        ("grid8", synth.grid(8, 8, 4, seed=1)), ("grid16", synth.grid(16, 16, 4, seed=1)),
        ("grid32", synth.grid(32, 32, 4, seed=1)), ("grid40", synth.grid(40, 40, 4, seed=1)),
        ("dag200", synth.random_dag(200, 4, 64, 4, seed=200)), ("dag1000", synth.random_dag(1000, 4, 64, 4, seed=1000)),
        ("mixed60", synth.random_dag(60, 3, 16, [2, 3, 4, 3, 2, 4, 4], seed=9)),
        ("mixed300", synth.random_dag(300, 3, 32, [2, 3, 4, 3, 2, 5], seed=4)),
        ("mixed2k", synth.random_dag(2000, 4, 64, [2, 3, 4, 3, 2, 4, 4], seed=9)),
        ("chain200", synth.grid(200, 1, 4, seed=5)), ("grid64", synth.grid(64, 64, 4, seed=1)),
        ("grid128", synth.grid(128, 128, 4, seed=1)), ("grid160", synth.grid(160, 160, 4, seed=1)),
        ("grid200", synth.grid(200, 200, 4, seed=1)), ("grid250", synth.grid(250, 250, 4, seed=1)),
        ("grid316", synth.grid(316, 316, 4, seed=2)),
        ("dag2p_3000", synth.random_dag(3000, 2, 64, 4, seed=3)), ("dag10k", synth.random_dag(10000, 4, 64, 4, seed=1)),
        ("dag3000", synth.random_dag(3000, 4, 64, 4, seed=5))]
if len(sys.argv) > 1:
    nets = [x for x in nets if x[0] in sys.argv[1:]]
out = {}
for name, mod in nets:
    with Engine(mod) as e:
        e.bp_set_evidence(Evidence.none())
        res = {}
        # 0: one launch per sweep; 1: resident tiles, grid barrier per sweep; 2: resident tiles, dataflow form
        for form in (2, 1, 0):
            e.set_option("multisweep", 2 if form else 0)
            e.set_option("flow", 1 if form == 2 else 0)
            for _ in range(3):
                r = e.bp_run_device(1e-6)
            reps = 20
            t0 = time.perf_counter()
            dev = 0.0
            for _ in range(reps):
                r = e.bp_run_device(1e-6)
                dev += e.bp_stats()["sweep_devclock_ms"]
            wall = (time.perf_counter() - t0) / reps
            res[form] = {"path": e.last_path(), "sweeps": r["sweeps"], "us_per_sweep_dev": dev / reps * 1e3 / r["sweeps"],
                         "us_per_run_wall": wall * 1e6, "beliefs": e.bp_beliefs(), "res": e.bp_residuals(),
                         "aborts": e.bp_stats()["resident_aborts"]}
        same = all(np.array_equal(res[0]["beliefs"], res[f]["beliefs"], equal_nan=True) and np.array_equal(res[0]["res"], res[f]["res"])
                   and res[0]["sweeps"] == res[f]["sweeps"] for f in (1, 2))
        lay = e.layout()
        row = {"nodes": mod.n, "tiles": lay["n_tiles"], "sweeps": res[0]["sweeps"], "same_bits": bool(same),
               "multi_path_taken": res[1]["path"], "aborts": res[2]["aborts"], "waves_per_block": e.info("resident_waves"),
               "per_sweep_launch": {k: round(res[0][k], 2) for k in ("us_per_sweep_dev", "us_per_run_wall")},
               "one_launch": {k: round(res[1][k], 2) for k in ("us_per_sweep_dev", "us_per_run_wall")},
               "one_launch_flow": {k: round(res[2][k], 2) for k in ("us_per_sweep_dev", "us_per_run_wall")}}
        out[name] = row
        print(name, json.dumps(row), flush=True)
if len(sys.argv) <= 1:
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "small_networks.json"), "w"), indent=1)

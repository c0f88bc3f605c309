# Small networks: one launch per sweep vs the whole run in one workgroup with the state in LDS (bn_small.hip); run on the GPU box.
# (the oracle is the checker here, nothing of it is timed)
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import Evidence, synth  # noqa: E402
from bayesiannetwork_amd.dsc import load_dsc  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402
import oracle  # noqa: E402

alarm, _ = load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))
nets = [("alarm_shaped", alarm), ("pearl", synth.pearl()), ("resume_chain", synth.resume_chain())]
for n, mp, seed in ((8, 2, 1), (12, 3, 2), (20, 3, 3), (27, 3, 4), (37, 4, 5), (48, 3, 6), (60, 3, 9), (80, 3, 10), (150, 3, 11)):
    nets.append((f"mixed{n}", synth.random_dag(n, mp, 16, [2, 3, 4, 3, 2, 4, 5], seed=seed)))
nets += [("grid8", synth.grid(8, 8, 4, seed=1)), ("grid12k3", synth.grid(12, 12, 3, seed=1)), ("dag60k4", synth.random_dag(60, 4, 16, 4, seed=5)),
         ("chain200", synth.grid(200, 1, 4, seed=5))]
if len(sys.argv) > 1:
    nets = [x for x in nets if x[0] in sys.argv[1:]]
for name, mod in nets:
    with Engine(mod) as e:
        lay = e.layout()
        ev = synth.random_evidence(mod, 0.05, seed=3)
        e.bp_set_evidence(ev)
        res = {}
        for form in (0, 1, 2):  # 0: one launch per sweep; 1: one workgroup, state in LDS (bn_small.hip); 2: what else multisweep = 2 gives
            e.set_option("multisweep", 0 if form == 0 else 2)
            e.set_option("small", 2 if form == 1 else 0)
            for _ in range(3):
                r = e.bp_run_device(1e-6)
            reps = 50
            t0 = time.perf_counter()
            dev = 0.0
            for _ in range(reps):
                r = e.bp_run_device(1e-6)
                dev += e.bp_stats()["sweep_devclock_ms"]
            wall = (time.perf_counter() - t0) / reps
            res[form] = {"path": e.last_path(), "sweeps": r["sweeps"], "us_sweep": round(dev / reps * 1e3 / r["sweeps"], 2),
                         "us_run_wall": round(wall * 1e6, 1), "beliefs": e.bp_beliefs(), "res": e.bp_residuals(), "msg": e.bp_messages()}
        o = oracle.bp_run(mod, ev, 1e-6)
        vs_oracle = float(np.nanmax(np.abs(res[1]["beliefs"] - o["beliefs"])))
        exact = np.array_equal(res[1]["beliefs"], o["beliefs"], equal_nan=True) and res[1]["sweeps"] == o["sweeps"]
        same = (np.array_equal(res[0]["beliefs"], res[1]["beliefs"], equal_nan=True) and np.array_equal(res[0]["res"], res[1]["res"])
                and res[0]["sweeps"] == res[1]["sweeps"] and all(np.array_equal(x, y, equal_nan=True) for x, y in zip(res[0]["msg"], res[1]["msg"])))
        print(name, "nodes", mod.n, "tiles", lay["n_tiles"], "small_waves", e.info("small_waves"), "lds", e.info("small_lds_bytes"),
              "same_bits_as_launches", same, "oracle_exact", exact, "max|d|", vs_oracle,
              {f: {k: res[f][k] for k in ("path", "sweeps", "us_sweep", "us_run_wall")} for f in res}, flush=True)

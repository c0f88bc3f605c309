import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.dsc import load_dsc
from bayesiannetwork_amd.engine import Engine
import oracle
alarm, _ = load_dsc("/root/repo/tests/golden/alarm_shaped.dsc")
evs = [synth.random_evidence(alarm, f, seed=q) for q, f in enumerate([0.0, 0.05, 0.1, 0.2] * 4)]
with Engine(alarm) as eng:
    for q, ev in enumerate(evs):
        o = oracle.bp_run(alarm, ev, 1e-9)
        eng.set_option("multisweep", 0)
        r0 = eng.bp_run(ev, 1e-9); h0 = eng.bp_residuals()
        eng.set_option("multisweep", 2)
        r3 = eng.bp_run(ev, 1e-9); h3 = eng.bp_residuals()
        print(q, "oracle", o["sweeps"], "launch", r0["sweeps"], eng.last_path(), "small", r3["sweeps"], "launch==oracle", np.array_equal(r0["beliefs"], o["beliefs"], equal_nan=True), "small==oracle", np.array_equal(r3["beliefs"], o["beliefs"], equal_nan=True),
              "hist launch", np.array_equal(h0, o["residuals"]), "hist small", np.array_equal(h3, o["residuals"]))
    eng.set_option("multisweep", 2)
    out = eng.bp_run_batch(evs, 1e-9)
    print("batch path", eng.last_path(), out["sweeps"].tolist())
    out = eng.bp_run_batch(evs[:5], 1e-3, 2)
    print("batch path", eng.last_path(), out["sweeps"].tolist())

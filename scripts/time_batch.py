import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.engine import Engine
for name, m in (("dag3000", synth.random_dag(3000, 4, 64, 4, seed=5)), ("grid128", synth.grid(128, 128, 4, seed=1)), ("grid64", synth.grid(64, 64, 4, seed=1))):
    with Engine(m) as e:
        for B in (1, 2, 4, 8):
            evs = [synth.random_evidence(m, 0.01, seed=7 + q) for q in range(B)]
            for mode in (0, 2):
                e.set_option("multisweep", mode)
                e.bp_set_evidence_batch(evs)
                for _ in range(3): r = e.bp_run_batch_device(1e-6)
                t0 = time.perf_counter(); sw = 0
                for _ in range(20):
                    r = e.bp_run_batch_device(1e-6); sw += int(r["sweeps"].sum())
                dt = time.perf_counter() - t0
                print(name, "B", B, "mode", mode, "path", e.last_path(), "us per set-sweep %.2f" % (dt / sw * 1e6), flush=True)

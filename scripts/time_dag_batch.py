#!/usr/bin/env python3
"""bn_bp_run_batch on BASELINE configs[1] (10 k-node DAG, register-resident path): us per set-sweep and edge-messages/s by batch
size; BN_DAG_SETS (environment) = sets that share a launch.  GPU box only:  python scripts/time_dag_batch.py [B ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16, 64]
g = synth.random_dag(10000, 4, 64, 4, seed=1)
with Engine(g) as eng:
    for B in sizes:
        evs = [synth.random_evidence(g, 0.01, seed=7 + q) for q in range(B)]
        eng.bp_set_evidence_batch(evs)
        for _ in range(3):
            out = eng.bp_run_batch_device(1e-3)
        reps = max(3, 64 // B)
        t0 = time.perf_counter()
        for _ in range(reps):
            out = eng.bp_run_batch_device(1e-3)
        dt = (time.perf_counter() - t0) / reps
        ss = float(sum(out["sweeps"]))
        print(f"sets per launch {os.environ.get('BN_DAG_SETS', '8')}  B={B:3d}  {dt * 1e6:8.1f} us per call  {dt * 1e6 / ss:6.2f} us per set-sweep  "
              f"{g.messages_per_sweep() * ss / dt / 1e9:6.2f} G edge-messages/s  path {eng.last_path()}  "
              f"in the kernels {eng.bp_stats()['sweep_devclock_ms'] * 1e3 / ss:5.2f} us per set-sweep")

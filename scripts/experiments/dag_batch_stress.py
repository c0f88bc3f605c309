#!/usr/bin/env python3
"""Repeats bn_bp_run_batch on the register-resident DAG path many times and compares every set's sweep count and bits with its single
run (the several-sets launch has no block barriers: a rare ordering slip would show as a differing bit or a bounded wait giving up).
GPU box:  python scripts/experiments/dag_batch_stress.py [calls]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for name, g in (("dag10k", synth.random_dag(10000, 4, 64, 4, seed=1)), ("dag600", synth.random_dag(600, 4, 48, 4, seed=43)),
                ("grid40", synth.grid(40, 40, 4, seed=5)), ("dag15k_stream", synth.random_dag(15000, 4, 64, 4, seed=3))):
    for B in (3, 8, 13):
        evs = [synth.random_evidence(g, 0.01 + 0.02 * (q % 5), seed=100 + q) for q in range(B)]
        with Engine(g) as eng:
            eng.set_option("dag", 2)
            singles = [eng.bp_run(ev, 1e-4) for ev in evs]
            eng.bp_set_evidence_batch(evs)
            t0 = time.perf_counter()
            n = max(10, calls // (4 if "stream" in name else 1))
            for c in range(n):
                out = eng.bp_run_batch_device(1e-4)
                if c % 10 == 0 or c == n - 1:
                    bel = eng.bp_beliefs_batch()
                    for q, r in enumerate(singles):
                        if out["sweeps"][q] != r["sweeps"] or not np.array_equal(bel[q], r["beliefs"]):
                            bad += 1
                            print("MISMATCH", name, B, c, q, out["sweeps"][q], r["sweeps"], flush=True)
            print(f"{name} B={B}: {n} calls, {(time.perf_counter() - t0) / n * 1e6:.0f} us per call, aborts {eng.info('dag_aborts')}, path {eng.last_path()}", flush=True)
            if eng.info("dag_aborts"):
                bad += 1
print("STRESS_OK" if bad == 0 else f"STRESS_FAILED {bad}")

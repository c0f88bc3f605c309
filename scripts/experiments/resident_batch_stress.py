#!/usr/bin/env python3
"""Repeats bn_bp_run_batch_device on the resident tiles (several sets per launch: no block barriers since round 5, waves count themselves
in, the first wave to need a verdict polls) and on the several-workgroup item kernel (chunks enqueued behind each other, one host wait)
many times and compares every set's sweep count and bits with its single run -- a rare ordering slip would show as a differing bit or a
bounded wait giving up.  GPU box:  python scripts/experiments/resident_batch_stress.py [calls]"""
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
cases = (("grid316", synth.grid(316, 316, 4, seed=2), {"multisweep": 2}, 2, 1e-3),
         ("grid64", synth.grid(64, 64, 4, seed=5), {"multisweep": 2, "mid": 0, "dag": 0}, 2, 1e-4),
         ("grid24", synth.grid(24, 24, 3, seed=8), {"multisweep": 2, "mid": 0, "small": 0, "dag": 0}, 2, 1e-6),
         ("mixed300", synth.random_dag(300, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=12), {}, 4, 1e-6))
for name, g, opts, path, eps in cases:
    for B in (2, 3, 4, 7, 16):
        evs = [synth.random_evidence(g, 0.01 + 0.02 * (q % 5), seed=100 + q) for q in range(B)]
        with Engine(g) as eng:
            for k, v in opts.items():
                eng.set_option(k, v)
            singles = [eng.bp_run(ev, eps) for ev in evs]
            eng.bp_set_evidence_batch(evs)
            t0 = time.perf_counter()
            n = max(10, calls // (8 if name == "grid316" else 1))
            for c in range(n):
                out = eng.bp_run_batch_device(eps)
                if c % 10 == 0 or c == n - 1:
                    bel = eng.bp_beliefs_batch()
                    for q, r in enumerate(singles):
                        if out["sweeps"][q] != r["sweeps"] or not np.array_equal(bel[q], r["beliefs"], equal_nan=True):
                            bad += 1
                            print("MISMATCH", name, B, c, q, out["sweeps"][q], r["sweeps"], flush=True)
            st = eng.bp_stats()
            print(f"{name} B={B}: {n} calls, {(time.perf_counter() - t0) / n * 1e6:.0f} us per call, path {eng.last_path()}, resident aborts {st['resident_aborts']}, "
                  f"mid aborts {eng.info('mid_aborts')}", flush=True)
            if eng.last_path() != path or st["resident_aborts"] or eng.info("mid_aborts"):
                bad += 1
print("STRESS_OK" if bad == 0 else f"STRESS_FAILED {bad}")

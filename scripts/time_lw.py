#!/usr/bin/env python3
"""Likelihood weighting on BASELINE configs[4] (10 k-node DAG, 1 % evidence): samples per second of bn_lw_run.
GPU box only:  python scripts/time_lw.py [samples_per_call]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
d = synth.random_dag(10000, 4, 64, 4, seed=1)
ev = synth.random_evidence(d, 0.01, seed=7).hard_states(d)
with Engine(d) as eng:
    for w in range(2):
        eng.lw_run(ev, n, seed=1, sample_begin=w * n)
    t0 = time.perf_counter()
    reps = 3
    for i in range(reps):
        eng.lw_run(ev, n, seed=1, sample_begin=(i + 2) * n)
    dt = time.perf_counter() - t0
print(f"{n} samples per call: {n * reps / dt / 1e6:.2f} M samples/s, {dt / reps * 1e3:.2f} ms per call")

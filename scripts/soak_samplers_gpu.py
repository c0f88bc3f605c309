#!/usr/bin/env python3
"""Randomised parity soak of the sampler family on the GPU box (not part of the test suite):  python scripts/soak_samplers_gpu.py [seconds] [seed]
Random DAGs (arities 2 ... 7, up to 6 parents: both sampling kernels) x random hard evidence x random sample counts, seeds and sample
offsets; per network: the sampled STATES bit-equal to the oracle's (likelihood_weighting.hpp:122-193 on the repository's stream), weights
<= 1e-12, the weighted histogram <= 1e-9 (fp64 atomics order); rejection sampling: counts, draws and acceptances exact; CPT fitting
from the sampled patterns bit-equal to the restatement."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (the checker)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
t_end = time.time() + budget
count = checks = 0
while time.time() < t_end:
    n = int(rng.choice([8, 40, 150, 600, 2500, 9000]))
    mp = int(rng.integers(1, 7))
    if rng.random() < 0.4:
        arities = [int(x) for x in rng.choice([2, 3, 4], size=int(rng.integers(1, 4)))]     # the straight-line kernel's domain (if tables <= 256 rows, <= 4 parents)
    else:
        arities = [int(x) for x in rng.choice([2, 3, 4, 5, 7], size=int(rng.integers(1, 4)))]
    g = synth.random_dag(n, mp, int(rng.choice([4, 16, 64])), arities if len(arities) > 1 else arities[0], seed=int(rng.integers(1, 1 << 30)))
    st = synth.random_evidence(g, float(rng.choice([0.0, 0.03, 0.15])), seed=int(rng.integers(1, 1 << 30))).hard_states(g)
    ns = int(rng.choice([64, 1000, 4096, 20000])) if n <= 2500 else int(rng.choice([64, 1000]))
    seed, begin = int(rng.integers(1, 1 << 40)), int(rng.choice([0, 7, 1 << 33]))
    with Engine(g) as e:
        hist = e.lw_run(st, ns, seed=seed, sample_begin=begin)
        small = e.info("lw_small")
        states, weights = e.lw_states(ns)
        o = oracle.lw_run(g, st, ns, seed=seed, s_begin=begin, states_cap=ns)
        assert np.array_equal(states, o["states"]), (g.name, "states", small)
        assert np.allclose(weights, o["weights"], rtol=1e-12, atol=0), (g.name, "weights")
        assert np.allclose(hist, o["hist"], rtol=1e-9, atol=1e-12), (g.name, "histogram")
        checks += 3
        if n <= 600:
            want = int(rng.choice([10, 200]))
            c, drawn, acc = e.rs_run(st, want, seed=seed, max_draw=1 << 16, sample_begin=begin)
            wc, wd, wa = oracle.rs_run(g, st, want, seed=seed, s_begin=begin, max_draw=1 << 16)
            assert (drawn, acc) == (wd, wa) and np.array_equal(c, wc), (g.name, "rejection sampling", drawn, wd, acc, wa)
            checks += 1
        if n <= 150 and ns >= 1000:
            pats, cnts = np.unique(states, axis=0, return_counts=True)
            assert np.array_equal(e.fit_cpt(pats, cnts), oracle.make_cpt(g, pats, cnts)), (g.name, "fit_cpt")
            checks += 1
    count += 1
    print(f"{count:4d} {g.name:28s} n={n:5d} max parents={mp} arities={arities} samples={ns} kernel={'straight-line' if small else 'generic'}", flush=True)
print(f"sampler soak ok: {count} networks, {checks} comparisons")

#!/usr/bin/env python3
"""Randomised soak of the IN-KERNEL halo exchange (the dataflow form across shard engines, bn_resident.hip) on one device:
    GPU_MAX_HW_QUEUES=16 python scripts/soak_shardflow_gpu.py [seconds] [seed]
n shard engines of a random network in ONE process, one thread per shard, their resident kernels co-resident, exchanging through peer
pointers (tests/shardflow_inproc.py's arrangement); each case three runs: sweep counts, residual histories, messages and marginals must
be bit-identical to the unsharded run.  Networks the resident kernel does not cover are skipped."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402
import shardflow_inproc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
t_end = time.time() + budget
count = skipped = 0
while time.time() < t_end:
    kind = int(rng.integers(0, 3))
    seed = int(rng.integers(1, 1 << 30))
    k = int(rng.choice([2, 3, 4]))
    if kind == 0:
        g = synth.grid(int(rng.integers(4, 70)), int(rng.integers(4, 70)), k, seed=seed)
    elif kind == 1:
        g = synth.random_dag(int(rng.integers(20, 1500)), 1, int(rng.integers(1, 6)), k, seed=seed)
    else:
        g = synth.random_dag(int(rng.choice([100, 600, 2500])), 2, int(rng.choice([8, 32])), k, seed=seed)
    nranks = int(rng.integers(2, 5))
    owner = None if rng.random() < 0.7 else rng.integers(0, nranks, size=g.n).astype(np.int32)
    ev = synth.random_evidence(g, float(rng.choice([0.0, 0.03, 0.2])), seed=int(rng.integers(1, 1 << 30)))
    eps = float(rng.choice([1e-3, 1e-6, 1e-9]))
    cap = int(rng.choice([0, 0, 5]))
    # eligibility: every rank's tiles must qualify for the resident kernel and fit the chip together
    probe = [Engine(g, rank=r, nranks=nranks, owner=owner) for r in range(nranks)]
    try:
        blobs = [s.peer_export() for s in probe]
        ok = all(s.peer_import(blobs) for s in probe)
    except Exception:  # noqa: BLE001
        ok = False
    finally:
        for s in probe:
            s.close()
    if not ok:
        skipped += 1
        continue
    shardflow_inproc.check(g, ev, eps, nranks, owner=owner, max_sweeps=cap)
    count += 1
    print(f"{count:4d} {g.name:24s} n={g.n:5d} k={k} ranks={nranks} owner={'random' if owner is not None else 'stripes'} eps={eps:g} cap={cap}", flush=True)
print(f"in-kernel exchange soak ok: {count} networks ({skipped} outside the resident kernel's domain skipped)")

#!/usr/bin/env python3
"""The in-kernel halo exchange with ONE PROCESS PER RANK (the deployment shape, minus the xGMI links): random grids, 2-4 worker processes
(tests/shardflow_worker.py) on device 0, peers mapped through hipIpc handles; three runs per case, every rank's sweep count, residual
history and marginals against the unsharded run.    python scripts/soak_shardflow_ipc_gpu.py [seconds] [seed]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t_end = time.time() + budget
count = 0
while time.time() < t_end:
    rows, cols = int(rng.integers(8, 200)), int(rng.integers(8, 200))
    nranks = int(rng.integers(2, 5))
    if rows < 2 * nranks:
        continue
    eps = float(rng.choice([1e-3, 1e-6, 0.0]))
    cap = int(rng.choice([1100, 40])) if eps == 0.0 else int(rng.choice([0, 0, 6]))
    if cap == 1100 and rows * cols > 6000:
        cap = 40
    g = synth.grid(rows, cols, 4, seed=rows * 31 + cols)   # (what the worker builds from rows, cols)
    ev = synth.random_evidence(g, 0.02, seed=3)
    with Engine(g) as one:
        for k in ("small", "mid", "dag"):
            one.set_option(k, 0)
        want = one.bp_run(ev, eps, cap)
        want_res = one.bp_residuals()
    reps = 3
    with tempfile.TemporaryDirectory() as work:
        env = dict(os.environ)
        if rows * cols > 40000:
            env["BN_RESIDENT_WAVES"] = "8"   # the ranks share ONE chip here: their blocks must be co-resident
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shardflow_worker.py"), str(r), str(nranks), work, str(rows), str(cols),
                                   repr(eps), str(cap), str(reps), "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                 for r in range(nranks)]
        logs = []
        for p in procs:
            try:
                logs.append(p.communicate(timeout=300)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()   # exactly the processes started here
                raise
        assert all(p.returncode == 0 for p in procs), "\n".join(l[-1500:] for l in logs)
        outs = [dict(np.load(os.path.join(work, f"out{r}.npz"))) for r in range(nranks)]
    if not all(bool(o["shard_flow"]) for o in outs):
        print(f"     {rows}x{cols} ranks={nranks}: in-kernel exchange not available (tiles beyond the chip at this size), skipped", flush=True)
        continue
    for i in range(reps):
        for o in outs:
            assert int(o["aborts"]) == 0 and int(o[f"path{i}"]) == 2 and int(o[f"flow{i}"]) == 1, (rows, cols, nranks, "path")
            assert int(o[f"sweeps{i}"]) == want["sweeps"] and np.array_equal(o[f"history{i}"], want_res), (rows, cols, nranks, "sweeps / history", i)
        bel = sum(o[f"beliefs{i}"] for o in outs)
        assert np.array_equal(bel, want["beliefs"], equal_nan=True), (rows, cols, nranks, "marginals", i)
    count += 1
    print(f"{count:4d} grid {rows}x{cols} ranks={nranks} eps={eps:g} cap={cap} sweeps={want['sweeps']}", flush=True)
print(f"in-kernel exchange over IPC, one process per rank: {count} cases ok")

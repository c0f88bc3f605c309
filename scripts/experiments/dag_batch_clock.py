# Phase stamps of the LAST set's turn in the sixth iteration of a batch on the register-resident DAG kernel (several sets per launch),
# every wave.  Needs the diagnostic library (scripts/experiments/build_dbg.sh) and BN_MI355X_LIB=build/libbn_dbg.so.
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import _lib, synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = synth.random_dag(10000, 4, 64, 4, seed=1)
L = _lib.lib()
with Engine(g) as e:
    evs = [synth.random_evidence(g, 0.01, seed=7 + q) for q in range(B)]
    e.bp_set_evidence_batch(evs)
    plan = e.dag_plan()
    nw = plan["blocks"] * 8
    zero = np.zeros((nw, 12), dtype=np.uint64)
    for _ in range(3):
        out = e.bp_run_batch_device(1e-3)
    assert e.last_path() == 5
    buf = np.zeros((nw, 12), dtype=np.uint64)
    assert L.bn_debug_dag_clock(buf.ctypes.data_as(ctypes.c_void_p), nw) == 0
    st = buf.astype(np.int64)
    ss = float(sum(out["sweeps"]))
    print(f"B = {B}: {e.bp_stats()['sweep_devclock_ms'] * 1e3 / ss:.2f} us per set-sweep in the kernels")
    ok = st[:, 1] != 0
    # 0 turn start, 1 verdict known, 9 inputs requested, [previous turn's arrival: 3 stores issued .. 4 drained .. 5 published], 2 inputs there, 6 turn done
    for nm, a_, b_ in (("wait verdict", 0, 1), ("request", 1, 9), ("previous arrival + inputs", 9, 2), ("arithmetic + stores issued", 2, 6), ("turn", 0, 6),
                       ("(drain of this turn's stores, in the next turn)", 3, 4), ("(arrival after the drain)", 4, 5)):
        d = (st[ok, b_] - st[ok, a_]) * 10
        print(f"   {nm:50s} ns median {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}  max {d.max():8.0f}")
    # the blocks that set the pace: the ones that do not wait
    kinds = np.full(nw, -1)
    cnt = np.diff(plan["slot_ptr"])
    has = cnt > 0
    kinds[has] = plan["tiles"][plan["slot_ptr"][:-1][has], 0]
    wait = ((st[:, 1] - st[:, 0]) * 10).reshape(-1, 8)
    blk_wait = np.median(wait, axis=1)
    order = np.argsort(blk_wait)
    print(" blocks by median wait (ns):", [(int(b), int(blk_wait[b])) for b in order[:6]], "...", [(int(b), int(blk_wait[b])) for b in order[-3:]])
    for b in order[:3]:
        print(f"  block {b}: per wave kind / wait / request / arrival+inputs / arithmetic / turn")
        for w in range(8):
            r = st[b * 8 + w]
            print(f"    kind {kinds[b * 8 + w]:2d}  {(r[1] - r[0]) * 10:6d} {(r[9] - r[1]) * 10:6d} {(r[2] - r[9]) * 10:6d} {(r[6] - r[2]) * 10:6d} {(r[6] - r[0]) * 10:6d}")

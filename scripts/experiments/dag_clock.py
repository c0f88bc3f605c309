# Phase stamps of one iteration (the sixth) of the register-resident DAG kernel, every wave, by tile kind.  Needs the diagnostic
# library:  bash scripts/experiments/build_dbg.sh   and   BN_MI355X_LIB=build/libbn_dbg.so python scripts/experiments/dag_clock.py
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import _lib, synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

n_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = synth.random_dag(n_nodes, 4, 64, 4, seed=1)
ev = synth.random_evidence(g, 0.01, seed=7)
L = _lib.lib()
with Engine(g) as e:
    e.set_option("dag", 2)
    FLOW = os.environ.get("DAGFLOW") == "1"   # the dataflow form (no grid barrier): stamps 0 start, 1 neighbours + verdict there, 2 inputs, 3 stores issued, 4 drained, 6 published
    e.set_option("dagflow", 1 if FLOW else 0)
    e.bp_set_evidence(ev)
    for _ in range(3):
        r = e.bp_run_device(1e-3)
    assert e.last_path() == 5
    plan = e.dag_plan()
    nw = plan["blocks"] * 8
    buf = np.zeros((nw, 12), dtype=np.uint64)
    rc = L.bn_debug_dag_clock(buf.ctypes.data_as(ctypes.c_void_p), nw)
    assert rc == 0
    st = buf.astype(np.int64)
    kinds = np.full(nw, -1)
    cnt = np.diff(plan["slot_ptr"])
    has = cnt > 0
    kinds[has] = plan["tiles"][plan["slot_ptr"][:-1][has], 0]
    ok = st[:, 1] != 0
    t0 = st[ok, 0].min()
    print(f"{n_nodes}-node DAG, {plan['n_tiles']} tiles on {plan['blocks']} blocks, {r['sweeps']} sweeps, {e.bp_stats()['sweep_devclock_ms'] * 1e3 / r['sweeps']:.2f} us per sweep")
    # stamps: 0 iteration start, 1 verdict known, 2 inputs arrived, 3 stores issued, 4 drained, 5 block synced, 6 published
    names = ["wait verdict", "loads", "compute + stores issued", "drain", "block sync", "publish"]
    order = [0, 1, 2, 3, 4, 5, 6]
    if FLOW:
        names = ["wait neighbours + verdict", "loads", "compute + stores issued", "drain", "publish", "-"]
        order = [0, 1, 2, 3, 4, 6, 6]
        print("dataflow form:", e.info("last_dag_flow"))
    for kind in (-1, 0, 1, 2, 3, 4, 5, 8):
        sel = ok & (kinds == kind)
        if not sel.any():
            continue
        s = st[sel]
        d = np.stack([s[:, order[i + 1]] - s[:, order[i]] for i in range(6)], axis=1) * 10
        print(f" kind {kind:2d} ({sel.sum()} waves): ns median / p90 / max")
        for i, nm in enumerate(names):
            print(f"   {nm:26s} {np.median(d[:, i]):8.0f} {np.percentile(d[:, i], 90):8.0f} {d[:, i].max():8.0f}")
        if kind in (3, 4, 5):
            for nm, a_, b_ in (("  contraction", 2, 7), ("  combine (shuffles)", 7, 8), ("  normalise + stores", 8, 3)):
                dd = (s[:, b_] - s[:, a_]) * 10
                print(f"   {nm:26s} {np.median(dd):8.0f} {np.percentile(dd, 90):8.0f} {dd.max():8.0f}")
        tot = (s[:, 6] - s[:, 0]) * 10
        print(f"   {'start -> published':26s} {np.median(tot):8.0f} {np.percentile(tot, 90):8.0f} {tot.max():8.0f}")
    print(" spread of verdict arrival over waves (ns):", (st[ok, 1].max() - st[ok, 1].min()) * 10, " first start -> last publish:", (st[ok, 6].max() - t0) * 10)

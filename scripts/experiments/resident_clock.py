# Phase stamps of one iteration (the sixth) of the resident kernel, every wave: needs the diagnostic library
#   (cd build/dbg_csrc && make EXTRA=-DBN_TILE_CLOCK OUT=../libbn_dbg.so)   and   BN_MI355X_LIB=build/libbn_dbg.so
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import _lib, synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 316
flow = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # 1: dataflow form (a wave waits for its neighbour tiles), 0: grid barrier
g = synth.grid(rows, rows, 4, seed=2)
ev = synth.random_evidence(g, 0.01, seed=7)
L = _lib.lib()
with Engine(g) as e:
    e.set_option("multisweep", 2)
    e.set_option("flow", flow)
    e.bp_set_evidence(ev)
    for _ in range(3):
        r = e.bp_run_device(1e-3)
    assert e.last_path() == 2
    n = 4096
    buf = np.zeros((n, 12), dtype=np.uint64)
    rc = L.bn_debug_tile_clock_resident(buf.ctypes.data_as(ctypes.c_void_p), n)
    assert rc == 0
    st = buf[buf[:, 1] != 0].astype(np.int64)
    if flow:
        st[:, 5] = st[:, 4]  # no block-level arrival in the dataflow form
    t0 = st[:, 0].min()
    names = ["wait (neighbours / verdict)" if flow else "wait verdict", "parent role", "contraction", "normalise+stores issued", "drain", "block sync", "granules"]
    # order of the stamps in time: 0 start, 1 verdict, 7 parent role, 8 contraction, 3 sweep issued, 4 drained, 5 synced, 6 published
    order = [0, 1, 7, 8, 3, 4, 5, 6]
    d = np.stack([st[:, order[i + 1]] - st[:, order[i]] for i in range(len(order) - 1)], axis=1) * 10  # ns
    print(f"{rows}x{rows} grid, {'dataflow form' if flow else 'grid barrier'}, {r['sweeps']} sweeps, {st.shape[0]} waves stamped; iteration 6, ns: median / p90 / max over waves")
    for i, nm in enumerate(names):
        print(f"  {nm:26s} {np.median(d[:, i]):8.0f} {np.percentile(d[:, i], 90):8.0f} {d[:, i].max():8.0f}")
    tot = (st[:, 6] - st[:, 0]) * 10
    print(f"  {'iteration (start->published)':26s} {np.median(tot):8.0f} {np.percentile(tot, 90):8.0f} {tot.max():8.0f}")
    print("  spread of iteration starts over waves (ns):", (st[:, 0].max() - st[:, 0].min()) * 10, " of verdict arrival:", (st[:, 1].max() - st[:, 1].min()) * 10)
    print("  first start -> last publish (ns):", (st[:, 6].max() - t0) * 10)

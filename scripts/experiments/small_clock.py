# Phase stamps of iteration 3 of the one-workgroup kernel (bn_small.hip), every wave: needs the diagnostic library
#   (cp bayesiannetwork_amd/csrc/* build/dbg_csrc/ && cd build/dbg_csrc && make EXTRA=-DBN_TILE_CLOCK OUT=../libbn_dbg.so)
#   and BN_MI355X_LIB=build/libbn_dbg.so
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import _lib, synth  # noqa: E402
from bayesiannetwork_amd.dsc import load_dsc  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "alarm"
if name == "alarm":
    g, _ = load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))
else:
    n, mp, seed = {"mixed20": (20, 3, 3), "mixed37": (37, 4, 5), "mixed60": (60, 3, 9)}[name]
    g = synth.random_dag(n, mp, 16, [2, 3, 4, 3, 2, 4, 5], seed=seed)
ev = synth.random_evidence(g, 0.05, seed=3)
L = _lib.lib()
with Engine(g) as e:
    e.bp_set_evidence(ev)
    for _ in range(3):
        r = e.bp_run_device(1e-6)
    assert e.last_path() == 3
    buf = np.zeros((16, 8), dtype=np.uint64)
    assert L.bn_debug_small_clock(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf[: e.info("small_waves")].astype(np.int64)
    names = ["entry items", "barrier 1", "run sums (round 0)", "normalise, stores", "product items", "residual", "barrier 2 + decision"]
    order = [0, 1, 2, 6, 7, 3, 4, 5]
    d = np.stack([st[:, order[i + 1]] - st[:, order[i]] for i in range(7)], axis=1) * 10
    print(f"{name}: {g.n} nodes, {e.info('small_waves')} waves, {r['sweeps']} sweeps, {e.bp_stats()['sweep_devclock_ms'] * 1e3 / r['sweeps']:.2f} us per sweep; iteration 3, ns: median / max over waves")
    for i, nm in enumerate(names):
        print(f"  {nm:30s} {np.median(d[:, i]):7.0f} {d[:, i].max():7.0f}")
    print("  per wave (ns):")
    for w in range(st.shape[0]):
        print("   ", w, d[w].tolist())

# Phase stamps of iteration 3 of the mid-size kernel (bn_mid.hip), thread 0 of every workgroup: needs the diagnostic library
#   (cp bayesiannetwork_amd/csrc/* build/dbg_csrc/ && cd build/dbg_csrc && make EXTRA=-DBN_TILE_CLOCK OUT=../libbn_dbg.so)
#   and BN_MI355X_LIB=build/libbn_dbg.so
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import _lib, synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

n, mp, seed = {"mixed80": (80, 3, 10), "mixed300": (300, 3, 12), "mixed1000": (1000, 3, 14), "mixed10k": (10000, 3, 20)}[sys.argv[1] if len(sys.argv) > 1 else "mixed300"]
g = synth.random_dag(n, mp, 16, [2, 3, 4, 3, 2, 4, 5], seed=seed)
ev = synth.random_evidence(g, 0.01 if n >= 10000 else 0.05, seed=3)
L = _lib.lib()
with Engine(g) as e:
    e.bp_set_evidence(ev)
    for _ in range(3):
        r = e.bp_run_device(1e-6)
    assert e.last_path() == 4
    buf = np.zeros((224, 8), dtype=np.uint64)
    assert L.bn_debug_mid_clock(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf[: e.info("mid_parts")].astype(np.int64)
    names = ["loads requested + entry items", "block barrier", "accumulator items", "product items", "residual", "grid barrier"]
    d = np.stack([st[:, i + 1] - st[:, i] for i in range(6)], axis=1) * 10
    print(f"{n} nodes, {e.info('mid_parts')} workgroups, {r['sweeps']} sweeps, {e.bp_stats()['sweep_devclock_ms'] * 1e3 / r['sweeps']:.2f} us per sweep; iteration 3, thread 0 of each workgroup, ns: median / max")
    for i, nm in enumerate(names):
        print(f"  {nm:32s} {np.median(d[:, i]):7.0f} {d[:, i].max():7.0f}")
    print("  start -> barrier released, ns: median", int(np.median(st[:, 6] - st[:, 0]) * 10), "max", int((st[:, 6] - st[:, 0]).max() * 10))
    print("  arrival at the grid barrier after the first workgroup, ns: median", int(np.median(st[:, 5] - st[:, 5].min()) * 10), "max", int((st[:, 5] - st[:, 5].min()).max() * 10))

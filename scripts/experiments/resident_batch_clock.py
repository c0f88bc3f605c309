# Phase stamps of the LAST set's turn in the sixth iteration of a batch on the resident kernel (several sets per launch), every wave.
# Needs the diagnostic library (scripts/experiments/build_dbg.sh) and BN_MI355X_LIB=build/libbn_dbg.so.
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import _lib, synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = synth.grid(316, 316, 4, seed=2)
L = _lib.lib()
with Engine(g) as e:
    evs = [synth.random_evidence(g, 0.01, seed=7 + q) for q in range(B)]
    e.bp_set_evidence_batch(evs)
    for _ in range(3):
        out = e.bp_run_batch_device(1e-3)
    assert e.last_path() == 2
    n = 4096
    buf = np.zeros((n, 12), dtype=np.uint64)
    assert L.bn_debug_tile_clock_resident(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
    st = buf[buf[:, 1] != 0].astype(np.int64)
    ss = float(sum(out["sweeps"]))
    print(f"B = {B}: {e.bp_stats()['sweep_devclock_ms'] * 1e3 / ss:.2f} us per set-sweep in the kernel, {st.shape[0]} waves stamped; ns: median / p90 / max over waves")
    names = ["wait verdict", "parent role", "contraction", "normalise + stores issued", "drain", "block sync", "granules"]
    order = [0, 1, 7, 8, 3, 4, 5, 6]
    d = np.stack([st[:, order[i + 1]] - st[:, order[i]] for i in range(len(order) - 1)], axis=1) * 10
    for i, nm in enumerate(names):
        print(f"  {nm:26s} {np.median(d[:, i]):8.0f} {np.percentile(d[:, i], 90):8.0f} {d[:, i].max():8.0f}")
    tot = (st[:, 6] - st[:, 0]) * 10
    print(f"  {'turn (start -> published)':26s} {np.median(tot):8.0f} {np.percentile(tot, 90):8.0f} {tot.max():8.0f}")

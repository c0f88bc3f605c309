# ALARM-shaped network: bn_bp_run_batch end to end (evidence arrays in, marginals on the host) vs its parts; run on the GPU box
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import _lib, synth  # noqa: E402
from bayesiannetwork_amd.dsc import load_dsc  # noqa: E402
from bayesiannetwork_amd.engine import Engine, _p  # noqa: E402

g, _ = load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))
L = _lib.lib()
with Engine(g) as e:
    for B in (16, 64):
        sets = [synth.random_evidence(g, 0.1, seed=100 + q) for q in range(B)]
        ne, node, off, val = e._pack_sets(sets)
        bel = np.empty((B, int(g.k.sum())))
        sweeps = np.zeros(B, dtype=np.int32)
        res = np.zeros(B)

        def call():
            _lib.check(L.bn_bp_run_batch(e._h, B, _p(ne, ctypes.c_int32), _p(node, ctypes.c_int32), _p(off, ctypes.c_int32), _p(val, ctypes.c_double),
                                         1e-6, 0, _p(bel, ctypes.c_double), _p(sweeps, ctypes.c_int32), _p(res, ctypes.c_double)))
        for _ in range(20):
            call()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            e.bp_set_evidence_batch(sets)
        dt_set = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            e.bp_run_batch_device(1e-6)
        dt_run = (time.perf_counter() - t0) / n
        print(f"B={B}: bn_bp_run_batch (arrays in -> marginals out) {dt * 1e6:.1f} us per call = {B / dt:.0f} queries/s; set_evidence_batch via Python {dt_set * 1e6:.1f} us, "
              f"run_batch_device {dt_run * 1e6:.1f} us, path {e.last_path()}", flush=True)

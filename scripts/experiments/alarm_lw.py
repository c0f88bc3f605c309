# ALARM-shaped network: likelihood weighting with the reference's default sample count (10 000) and more; run on the GPU box
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd.dsc import load_dsc  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

g, _ = load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))
ev = np.full(g.n, -1, dtype=np.int32)
ev[[3, 17]] = 0
with Engine(g) as e:
    for n in (10000, 100000, 1000000):
        for _ in range(3):
            e.lw_run(ev, n, seed=1)
        reps = 20
        t0 = time.perf_counter()
        for i in range(reps):
            e.lw_run(ev, n, seed=1, sample_begin=i * n)
        dt = (time.perf_counter() - t0) / reps
        print(f"{n} samples: {dt * 1e6:.1f} us per call = {n / dt:.3g} samples/s", flush=True)

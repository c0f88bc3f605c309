import sys, time, numpy as np
sys.path.insert(0, '.')
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.engine import Engine
g = synth.grid(316, 316, 4, seed=2)
evs = [synth.random_evidence(g, 0.01, seed=7 + q) for q in range(8)]
with Engine(g) as eng:
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        eng.bp_run_view(evs[0], 1e-3)
    ts = []
    for i in range(400):
        a = time.perf_counter(); eng.bp_run_view(evs[i % 8], 1e-3); ts.append(time.perf_counter() - a)
    ts = np.array(ts) * 1e3
    print("per-call ms: median %.4f mean %.4f p10 %.4f p90 %.4f max %.4f" % (np.median(ts), ts.mean(), np.percentile(ts, 10), np.percentile(ts, 90), ts.max()))
    for k in range(0, 400, 40): print("  block of 40: mean %.4f" % ts[k:k+40].mean())

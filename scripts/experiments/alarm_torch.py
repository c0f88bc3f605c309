import os, sys, time
sys.path.insert(0, "/root/repo")
mode = sys.argv[1]
if mode != "none":
    import torch
    if mode == "setdev":
        torch.cuda.set_device(0)
    if mode == "sync":
        torch.cuda.set_device(0); torch.cuda.synchronize()
    if mode == "alloc":
        torch.cuda.set_device(0); x = torch.zeros(1024, device="cuda"); torch.cuda.synchronize()
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.dsc import load_dsc
from bayesiannetwork_amd.engine import Engine
g, _ = load_dsc("/root/repo/tests/golden/alarm_shaped.dsc")
evs = [synth.random_evidence(g, 0.1, seed=7 + q) for q in range(8)]
def loop(f, n=400):
    for ev in evs[:2]: f(ev)
    t0 = time.perf_counter()
    for i in range(n): f(evs[i % 8])
    return (time.perf_counter() - t0) / n * 1e6
with Engine(g) as e:
    a = loop(lambda ev: e.bp_run_view(ev, 1e-6)); b = loop(lambda ev: e.bp_run(ev, 1e-6)); c = loop(lambda ev: e.bp_run_view(ev, 1e-6))
    print(mode, f"run_view {a:.1f} us, run {b:.1f} us, run_view again {c:.1f}", flush=True)

"""bn_create on the ALARM-shaped network while earlier engines stay ALIVE (what tests/cpp/bench_dropin.cpp measures): BN_CREATE_TIMING=1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesiannetwork_amd.dsc import load_dsc
from bayesiannetwork_amd.engine import Engine
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
g = load_dsc(os.path.join(root, "tests/golden/alarm_shaped.dsc"))[0]
keep = []
for rep in range(5):
    print(f"== alarm engine {rep} (earlier ones alive)", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    e = Engine(g, device=0)
    dt = time.perf_counter() - t0
    print(f"== {dt * 1e3:.2f} ms", {k: e.info('create_us_' + k) / 1e3 for k in ("plan", "small", "mid", "dag", "device")}, file=sys.stderr, flush=True)
    keep.append(e)

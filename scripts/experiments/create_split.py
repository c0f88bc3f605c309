"""Where bn_create's device side goes (BN_CREATE_TIMING=1: one stderr line per step), on the three bench networks; first and third engine of the process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.dsc import load_dsc
from bayesiannetwork_amd.engine import Engine
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nets = {"alarm": load_dsc(os.path.join(root, "tests/golden/alarm_shaped.dsc"))[0], "dag10k": synth.random_dag(10000, 4, 64, 4, seed=1),
        "grid316": synth.grid(316, 316, 4, seed=2)}
for name, g in nets.items():
    for rep in range(3):
        print(f"== {name} engine {rep}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        e = Engine(g, device=0)
        dt = time.perf_counter() - t0
        print(f"== {name} engine {rep}: {dt * 1e3:.2f} ms", {k: e.info('create_us_' + k) / 1e3 for k in ("plan", "small", "mid", "dag", "device")}, file=sys.stderr, flush=True)
        e.close()

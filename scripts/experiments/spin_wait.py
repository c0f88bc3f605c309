"""Host's wait for a run: hipStreamSynchronize (BN_SPIN_US unset / 0) against a spin on hipStreamQuery (BN_SPIN_US=300), per process.
us per query, evidence staged (bn_bp_run_device) and host to host (bn_bp_run_view), on the ALARM-shaped net, configs[1], configs[2]."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.dsc import load_dsc
from bayesiannetwork_amd.engine import Engine
nets = {"alarm": (load_dsc(os.path.join(ROOT, "tests/golden/alarm_shaped.dsc"))[0], 0.1, 1e-6), "dag10k": (synth.random_dag(10000, 4, 64, 4, seed=1), 0.01, 1e-3),
        "grid316": (synth.grid(316, 316, 4, seed=2), 0.01, 1e-3)}
out = {"BN_SPIN_US": os.environ.get("BN_SPIN_US")}
for name, (g, frac, eps) in nets.items():
    evs = [synth.random_evidence(g, frac, seed=7 + q) for q in range(8)]
    with Engine(g) as e:
        t_end = time.time() + 0.3
        while time.time() < t_end:
            e.bp_run_view(evs[0], eps)
        e.bp_set_evidence(evs[0])
        dev = []
        for i in range(300):
            t0 = time.perf_counter(); e.bp_run_device(eps); dev.append(time.perf_counter() - t0)
        h2h = []
        for i in range(300):
            t0 = time.perf_counter(); e.bp_run_view(evs[i % 8], eps); h2h.append(time.perf_counter() - t0)
        dev.sort(); h2h.sort()
        out[name] = {"run_device_us": round(dev[150] * 1e6, 2), "run_view_us": round(h2h[150] * 1e6, 2), "kernel_us": round(e.bp_stats()["sweep_devclock_ms"] * 1e3, 2)}
print(json.dumps(out))

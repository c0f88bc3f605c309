# ALARM-shaped network: wall time per query of the host entry points, in both orders (run on the GPU box)
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.dsc import load_dsc  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

g, _ = load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))
evs = [synth.random_evidence(g, 0.1, seed=7 + q) for q in range(8)]


def loop(f, n=400):
    for ev in evs[:2]:
        f(ev)
    t0 = time.perf_counter()
    for i in range(n):
        f(evs[i % 8])
    return (time.perf_counter() - t0) / n * 1e6


with Engine(g) as e:
    for rep in range(3):
        a = loop(lambda ev: e.bp_run_view(ev, 1e-6))
        b = loop(lambda ev: e.bp_run(ev, 1e-6))
        e.set_option("beliefs_direct", 0)
        c = loop(lambda ev: e.bp_run_view(ev, 1e-6))
        e.set_option("beliefs_direct", 1)

        def staged(ev):
            e.bp_set_evidence(ev)
            e.bp_run_device(1e-6)
        d = loop(staged)
        print(f"rep {rep}: run_view {a:.1f} us, run {b:.1f} us, run_view without direct beliefs {c:.1f} us, set_evidence + run_device {d:.1f} us", flush=True)

# Three engines (one-workgroup, several-workgroup, resident-tile paths) queried from three host threads at once; run on the GPU box
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.dsc import load_dsc  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402
import oracle  # noqa: E402

alarm, _ = load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))
nets = [("small", alarm), ("mid", synth.random_dag(300, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=12)), ("resident", synth.grid(64, 64, 4, seed=1)),
        ("mid_batch", synth.random_dag(150, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=11)),
        ("mid_wide", synth.random_dag(2000, 4, 64, [2, 3, 4, 3, 2, 4, 4], seed=9)), ("resident_direct", synth.grid(316, 316, 4, seed=2))]   # ~150 / 196 co-resident workgroups each
bad = []


def work(name, g):
    evs = [synth.random_evidence(g, 0.05, seed=q) for q in range(4)]
    eps = 1e-3 if name == "resident_direct" else 1e-6
    want = [oracle.bp_run(g, ev, eps) for ev in evs]
    with Engine(g) as e:
        for i in range(150):
            if name == "mid_batch":
                out = e.bp_run_batch(evs * 8, eps)
                ok = all(np.array_equal(out["beliefs"][q], want[q % 4]["beliefs"]) for q in range(32))
            else:
                r = e.bp_run_view(evs[i % 4], eps)
                ok = r["sweeps"] == want[i % 4]["sweeps"] and np.array_equal(r["beliefs"], want[i % 4]["beliefs"])
            if not ok:
                bad.append((name, i))
                return
        print(name, "path", e.last_path(), "aborts", e.info("mid_aborts"), e.bp_stats()["resident_aborts"], flush=True)


ts = [threading.Thread(target=work, args=x) for x in nets]
for t in ts:
    t.start()
for t in ts:
    t.join()
print("mismatches:", bad)

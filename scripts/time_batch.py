# Batched evidence sets (bn_bp_run_batch_device): microseconds per set-sweep by batch size, on the per-sweep
# launches (mode 0: one evidence set per blockIdx.y) and, where eligible, the resident kernel (mode 2).
# Run on the GPU box:  python scripts/time_batch.py [network ...]
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.dsc import load_dsc  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

alarm, _ = load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))
nets = [("alarm_shaped", alarm), ("dag10k", synth.random_dag(10000, 4, 64, 4, seed=1)), ("dag3000", synth.random_dag(3000, 4, 64, 4, seed=5)),
        ("mixed300", synth.random_dag(300, 3, 32, [2, 3, 4, 3, 2, 5], seed=4)), ("grid64", synth.grid(64, 64, 4, seed=1)),
        ("grid128", synth.grid(128, 128, 4, seed=1)), ("grid200", synth.grid(200, 200, 4, seed=1)),
        ("grid250", synth.grid(250, 250, 4, seed=1)), ("grid316", synth.grid(316, 316, 4, seed=2))]
if len(sys.argv) > 1:
    nets = [x for x in nets if x[0] in sys.argv[1:]]
out = {}
for name, m in nets:
    with Engine(m) as e:
        for mode in (0, 2):
            for B in (1, 2, 4, 8, 16, 32, 64):
                if m.n > 50000 and B > 16:
                    continue
                evs = [synth.random_evidence(m, 0.01, seed=7 + q) for q in range(B)]
                e.set_option("multisweep", mode)
                e.bp_set_evidence_batch(evs)
                for _ in range(3):
                    r = e.bp_run_batch_device(1e-6)
                if mode == 2 and e.last_path() != 2:
                    break
                reps = 10
                t0 = time.perf_counter()
                sw = 0
                for _ in range(reps):
                    r = e.bp_run_batch_device(1e-6)
                    sw += int(r["sweeps"].sum())
                dt = time.perf_counter() - t0
                row = {"path": e.last_path(), "us_per_set_sweep": round(dt / sw * 1e6, 3), "ms_per_call": round(dt / reps * 1e3, 4),
                       "msgs_per_s": m.messages_per_sweep() * sw / dt}
                out[f"{name}_mode{mode}_B{B}"] = row
                print(name, "mode", mode, "B", B, json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "batch_times.json"), "w"), indent=1)

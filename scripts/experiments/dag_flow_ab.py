"""The register-resident DAG path's single query under the grid barrier ("dagflow" 0) and in its dataflow form ("dagflow" 1):
us per sweep (device clock: first sweep's start -> the outcome's report), us per query (host, evidence staged), sweep counts,
and whether the two give the same bits.  GPU box:  python scripts/experiments/dag_flow_ab.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.engine import Engine


def measure(eng, evs, eps, reps=60):
    for i in range(8):
        eng.bp_set_evidence(evs[i % len(evs)])
        eng.bp_run_device(eps)
    dev, sweeps, dt = 0.0, 0, 0.0
    for i in range(reps):
        eng.bp_set_evidence(evs[i % len(evs)])
        t0 = time.perf_counter()
        r = eng.bp_run_device(eps)
        dt += time.perf_counter() - t0
        dev += eng.bp_stats()["sweep_devclock_ms"]
        sweeps += r["sweeps"]
    return {"path": eng.last_path(), "flow": eng.info("last_dag_flow"), "us_per_sweep": round(dev / sweeps * 1e3, 3),
            "us_per_query": round(dt / reps * 1e6, 2), "kernel_us_per_query": round(dev / reps * 1e3, 2), "sweeps": sweeps / reps}


nets = [("dag10k (configs[1])", synth.random_dag(10000, 4, 64, 4, seed=1), 1e-3),
        ("dag10k eps 1e-6", synth.random_dag(10000, 4, 64, 4, seed=1), 1e-6),
        ("mixed-arity 10k", synth.random_dag(10000, 4, 64, [2, 3, 4], seed=8), 1e-3),
        ("dag3000", synth.random_dag(3000, 4, 64, 4, seed=8), 1e-3),
        ("dag1000", synth.random_dag(1000, 4, 64, 4, seed=9), 1e-3),
        ("grid64", synth.grid(64, 64, 4, seed=5), 1e-3),
        ("grid100", synth.grid(100, 100, 4, seed=5), 1e-3)]
for name, g, eps in nets:
    evs = [synth.random_evidence(g, 0.01, seed=7 + q) for q in range(8)]
    row = {}
    with Engine(g) as eng:
        eng.set_option("dag", 2)
        bel = {}
        for flow in (0, 1, 0, 1):
            eng.set_option("dagflow", flow)
            row[f"flow{flow}" + ("_again" if f"flow{flow}" in row else "")] = measure(eng, evs, eps)
            r = eng.bp_run(evs[0], eps)
            bel.setdefault(flow, (r["sweeps"], r["beliefs"].copy()))
        row["same_bits"] = bool(bel[0][0] == bel[1][0] and np.array_equal(bel[0][1], bel[1][1], equal_nan=True))
        row["info"] = {k: eng.info(k) for k in ("dag_blocks", "dag_tiles", "dag_flow_eligible", "dag_flow_max_nbr", "dag_aborts")}
    print(name, json.dumps(row), flush=True)

"""dag10k, dataflow form: us per executed iteration for one setting of BN_DAG_FLOW_SLEEP (read once per process)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.engine import Engine
g = synth.random_dag(10000, 4, 64, 4, seed=1)
evs = [synth.random_evidence(g, 0.01, seed=7 + q) for q in range(8)]
with Engine(g) as eng:
    for flow in (1, 0):
        eng.set_option("dagflow", flow)
        for i in range(8):
            eng.bp_set_evidence(evs[i % 8]); eng.bp_run_device(1e-3)
        dev = sw = 0
        for i in range(60):
            eng.bp_set_evidence(evs[i % 8]); r = eng.bp_run_device(1e-3)
            dev += eng.bp_stats()["sweep_devclock_ms"]; sw += r["sweeps"]
        its = sw + (60 if flow else 0)
        print(json.dumps({"sleep": os.environ.get("BN_DAG_FLOW_SLEEP"), "flow": eng.info("last_dag_flow"), "kernel_us_per_query": round(dev / 60 * 1e3, 2),
                          "us_per_sweep": round(dev / sw * 1e3, 3), "us_per_executed_iteration": round(dev / its * 1e3, 3)}), flush=True)

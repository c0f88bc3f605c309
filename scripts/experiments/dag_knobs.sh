#!/bin/bash
# config 2 on the register-resident DAG path under experiment switches (blocks, poll pause): us per sweep
for cap in 224 192; do for sl in 0 5 10 15 20 30 40 60; do
  echo -n "cap=$cap delay=$sl  "
  BN_DAG_CAP=$cap BN_DAG_DELAY=$sl python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.engine import Engine
g = synth.random_dag(10000, 4, 64, 4, seed=1)
ev = synth.random_evidence(g, 0.01, seed=7)
with Engine(g) as e:
    e.bp_set_evidence(ev)
    for _ in range(5): e.bp_run_device(1e-3)
    dev = sw = 0
    for _ in range(40):
        r = e.bp_run_device(1e-3); dev += e.bp_stats()["sweep_devclock_ms"]; sw += r["sweeps"]
    print(e.info("dag_blocks"), "blocks", round(dev / sw * 1e3, 2), "us per sweep, path", e.last_path())
PY
done; done

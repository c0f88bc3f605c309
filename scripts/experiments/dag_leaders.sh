#!/bin/bash
# config 2 on the register-resident DAG path: every block collects the barrier granules (leaders=0) or every L-th does (BN_DAG_LEADERS=L,
# the others poll its line; BN_DAG_HOP = extra ticks before a follower's first poll): us per sweep
run() {
  python - <<'PY'
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
    for _ in range(60):
        r = e.bp_run_device(1e-3); dev += e.bp_stats()["sweep_devclock_ms"]; sw += r["sweeps"]
    print(round(dev / sw * 1e3, 2), "us per sweep, path", e.last_path(), "aborts", e.info("dag_aborts"))
PY
}
for rep in 1 2; do
  echo -n "rep=$rep leaders=0         "; BN_DAG_LEADERS=0 run
  for L in 4 8 16; do for hop in 0 60 120; do
    echo -n "rep=$rep leaders=$L hop=$hop  "; BN_DAG_LEADERS=$L BN_DAG_HOP=$hop run
  done; done
done

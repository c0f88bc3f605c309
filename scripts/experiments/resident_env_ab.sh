#!/bin/bash
# resident tiles: A/B of an environment switch in one process sequence on one box.
#   bash scripts/experiments/resident_env_ab.sh BN_RESIDENT_EARLY "0 1" "316 250 128"
VAR=$1; VALS=${2:-0 1}; ROWS=${3:-316}
run() {
  python - $1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.engine import Engine
rows = int(sys.argv[1])
g = synth.grid(rows, rows, 4, seed=2)
ev = synth.random_evidence(g, 0.01, seed=7)
with Engine(g) as e:
    e.set_option("dag", 0); e.set_option("mid", 0)
    e.bp_set_evidence(ev)
    for _ in range(5): e.bp_run_device(1e-3)
    dev = sw = 0
    for _ in range(60):
        r = e.bp_run_device(1e-3); dev += e.bp_stats()["sweep_devclock_ms"]; sw += r["sweeps"]
    print(round(dev / sw * 1e3, 3), "us per sweep, path", e.last_path(), "sweeps", r["sweeps"])
PY
}
for rep in 1 2 3; do for rows in $ROWS; do for v in $VALS; do
  echo -n "rep=$rep $VAR=$v rows=$rows  "; export $VAR=$v; run $rows
done; done; done

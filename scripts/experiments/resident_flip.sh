#!/bin/bash
# resident tiles, 8-wave blocks: the two waves of a SIMD take an iteration's roles in the same (0) or in opposite (1) order;
# "prev" = the library of the commit before (build/libbn_prev.so), when it is there
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
    e.bp_set_evidence(ev)
    for _ in range(5): e.bp_run_device(1e-3)
    dev = sw = 0
    for _ in range(60):
        r = e.bp_run_device(1e-3); dev += e.bp_stats()["sweep_devclock_ms"]; sw += r["sweeps"]
    print(round(dev / sw * 1e3, 3), "us per sweep, path", e.last_path(), "sweeps", r["sweeps"])
PY
}
for rep in 1 2 3; do for rows in ${1:-316}; do
  if [ -f build/libbn_prev.so ]; then echo -n "rep=$rep prev   rows=$rows  "; BN_MI355X_LIB=build/libbn_prev.so run $rows; fi
  for f in 0 1; do echo -n "rep=$rep flip=$f rows=$rows  "; BN_RESIDENT_FLIP=$f run $rows; done
done; done

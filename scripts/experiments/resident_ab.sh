#!/bin/bash
# resident tiles: first poll from the block's own arrival on (-1) against the predicted arrival of the last block + margin (0, 10, 20)
for rep in 1 2; do for d in -1 0 10 20 -1 0; do
  echo -n "rep=$rep margin=$d  "
  BN_RESIDENT_DELAY=$d python - 316 <<'PY'
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
    print(round(dev / sw * 1e3, 3), "us per sweep, path", e.last_path())
PY
done; done

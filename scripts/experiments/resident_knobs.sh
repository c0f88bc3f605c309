#!/bin/bash
# resident tiles (bn_resident.hip, direct form): margin between the predicted arrival of the last block and a block's first poll
for rows in 316 200 128 64; do for d in 0 10 20 30 40 60 80 120; do
  echo -n "grid=$rows margin=$d  "
  BN_RESIDENT_DELAY=$d python - $rows <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.engine import Engine
rows = int(sys.argv[1])
g = synth.grid(rows, rows, 4, seed=2)
ev = synth.random_evidence(g, 0.01, seed=7)
with Engine(g) as e:
    e.set_option("multisweep", 2); e.set_option("mid", 0); e.set_option("dag", 0)
    e.bp_set_evidence(ev)
    for _ in range(5): e.bp_run_device(1e-3)
    dev = sw = 0
    for _ in range(40):
        r = e.bp_run_device(1e-3); dev += e.bp_stats()["sweep_devclock_ms"]; sw += r["sweeps"]
    print(e.info("resident_blocks"), "blocks x", e.info("resident_waves"), "waves", round(dev / sw * 1e3, 2), "us per sweep, path", e.last_path())
PY
done; done

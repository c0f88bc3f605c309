#!/bin/bash
# register-resident DAG path, single queries: this library against build/libbn_prev.so (the commit before), same box, alternating
for rep in 1 2 3 4; do for lib in prev cur; do
  if [ $lib = prev ]; then export BN_MI355X_LIB=build/libbn_prev.so; else unset BN_MI355X_LIB; fi
  python - $lib <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.engine import Engine
nets = [("dag10k", synth.random_dag(10000, 4, 64, 4, seed=1)), ("dag3000", synth.random_dag(3000, 4, 64, 4, seed=8)),
        ("dag1000", synth.random_dag(1000, 4, 64, 4, seed=9)), ("dag300", synth.random_dag(300, 4, 32, 4, seed=5)), ("grid64", synth.grid(64, 64, 4, seed=5))]
out = []
for name, g in nets:
    ev = synth.random_evidence(g, 0.01, seed=7)
    with Engine(g) as e:
        e.set_option("dag", 2)
        e.bp_set_evidence(ev)
        for _ in range(5): e.bp_run_device(1e-3)
        dev = sw = 0
        for _ in range(150):
            r = e.bp_run_device(1e-3); dev += e.bp_stats()["sweep_devclock_ms"]; sw += r["sweeps"]
        out.append(f"{name} {dev / sw * 1e3:.2f}")
print(sys.argv[1], "  ".join(out))
PY
done; done

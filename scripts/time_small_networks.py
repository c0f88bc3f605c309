# Per-sweep kernel time on small / mixed-arity networks (the any-arity tile variant); run on the GPU box.
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from bayesiannetwork_amd.engine import Engine
from bayesiannetwork_amd.dsc import load_dsc
from bayesiannetwork_amd import synth, Evidence
m, names = load_dsc('/root/repo/tests/golden/alarm_shaped.dsc')
for name, mod in [('alarm', m), ('mixed2k', synth.random_dag(2000, 4, 64, [2,3,4,3,2,4,4], seed=9)), ('k3_m3_2k', synth.random_dag(2000, 3, 64, 3, seed=3)), ('k5_2k', synth.random_dag(2000, 3, 64, 5, seed=3))]:
    with Engine(mod) as e:
        e.bp_set_evidence(Evidence.none())
        for _ in range(3): r = e.bp_run_device(1e-6)
        st = e.bp_stats()
        cls = e.layout_classes()
        print(name, mod.n, 'sweeps', r['sweeps'], 'us/sweep', st['sweep_kernel_ms']*1e3/max(st['sweep_launches'],1), 'variants', sorted(set(c['variant'] for c in cls)), 'generic nodes', sum(c['n_nodes'] for c in cls if c['variant']==0))

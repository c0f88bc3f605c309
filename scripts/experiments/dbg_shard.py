# n shard engines of a grid in ONE process (threads), in-kernel exchange, compared with the unsharded engine message by message:
#   GPU_MAX_HW_QUEUES=16 python scripts/experiments/dbg_shard.py ROWS COLS [NRANKS]
import os
import sys
import threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.engine import Engine
R, C = int(sys.argv[1]), int(sys.argv[2])
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g = synth.grid(R, C, 4, seed=R * 31 + C)
ev = synth.random_evidence(g, 0.02, seed=3)
for cap in (1, 2):
    with Engine(g) as one:
        want = one.bp_run(ev, 1e-6, cap)
        wres = one.bp_residuals(); wpi, wlam = one.bp_messages()
    sh = [Engine(g, rank=r, nranks=N) for r in range(N)]
    blobs = [s.peer_export() for s in sh]
    print("import", [s.peer_import(blobs) for s in sh])
    for s in sh: s.bp_set_evidence(ev)
    outs = [None] * N
    def work(i):
        try:
            outs[i] = sh[i].bp_run_device(1e-6, cap)
        except Exception as ex:
            outs[i] = {'sweeps': -1, 'err': str(ex)[:80]}
    for s_ in sh: print('  rank', s_.rank, 'tiles', s_.layout()['n_tiles'], 'interior', s_.layout()['n_interior_tiles'], 'nbr_max', s_.info('nbr_max'), 'pub', np.unique(s_.flow_tables()[1]))
    th = [threading.Thread(target=work, args=(i,)) for i in range(N)]
    [t.start() for t in th]; [t.join() for t in th]
    print("cap", cap, "sweeps", [o["sweeps"] for o in outs], want["sweeps"])
    print(" want res", wres[:5]); 
    print(" outs", outs)
    if any(o["sweeps"] < 0 for o in outs):
        for s in sh: s.close()
        continue
    for s in sh: print(" got res ", s.bp_residuals()[:5], "aborts", s.bp_stats()["resident_aborts"])
    moff = g.msg_off
    child = np.repeat(np.arange(g.n), np.diff(g.in_ptr))
    for s in sh:
        pi, lam = s.bp_messages(); rpi, rlam = s.edge_refs()
        seen = np.repeat(rpi >= 0, np.diff(moff))
        badpi = np.nonzero(seen & (pi != wpi))[0]; badlam = np.nonzero(seen & (lam != wlam))[0]
        print(" rank", s.rank, "bad pi", badpi.size, "bad lam", badlam.size)
        if badpi.size:
            e = np.searchsorted(moff, badpi[0], side='right') - 1
            print("   first bad pi edge", e, "parent", g.in_idx[e], "child", child[e], "cut?", rlam[e] < 0, pi[moff[e]:moff[e+1]], wpi[moff[e]:moff[e+1]])
        if badlam.size:
            e = np.searchsorted(moff, badlam[0], side='right') - 1
            print("   first bad lam edge", e, "parent", g.in_idx[e], "child", child[e], "cut?", rlam[e] < 0, lam[moff[e]:moff[e+1]], wlam[moff[e]:moff[e+1]])
    bel = sum(s.bp_beliefs() for s in sh)
    print(" beliefs equal", np.array_equal(bel, want["beliefs"]))
    for s in sh: s.close()

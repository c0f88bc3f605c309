"""Neighbour tiles of the register-resident DAG plan's tiles (who a tile would wait for in a dataflow form of bp_dag_kernel):
child tile C <-> the parent tiles that hold the items of its nodes and of its nodes' parents; symmetric.  Host-only engine."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.engine import Engine

def stats(name, g):
    with Engine(g, device=-2) as e:
        p = e.dag_plan()
    if p is None:
        print(name, "not eligible"); return
    nt = p["n_tiles"]
    tiles = p["tiles"]
    ctile = np.full(g.n, -1)
    ptiles = [set() for _ in range(g.n)]
    for t in range(nt):
        kind = tiles[t, 0]; lb = tiles[t, 2]
        if kind < 8:
            v = p["cnode"][lb:lb + 64, 0]
            ctile[v[v >= 0]] = t
        else:
            v = p["pitem"][lb:lb + 64, 0]
            for u in v[v >= 0]:
                ptiles[u].add(t)
    nbr = [set() for _ in range(nt)]
    for v in range(g.n):
        c = ctile[v]
        for t in ptiles[v]:
            nbr[c].add(t); nbr[t].add(c)
        for e_ in range(g.in_ptr[v], g.in_ptr[v + 1]):
            u = g.in_idx[e_]
            for t in ptiles[u]:
                nbr[c].add(t); nbr[t].add(c)
    cnt = np.array([len(s) for s in nbr])
    kinds = tiles[:, 0]
    print(f"{name}: {nt} tiles on {p['blocks']} blocks (stream {p['stream']}), neighbours per tile: max {cnt.max()}, mean {cnt.mean():.1f}, "
          f"p90 {np.percentile(cnt, 90):.0f}; >16: {(cnt > 16).sum()}, >32: {(cnt > 32).sum()}, >64: {(cnt > 64).sum()}")
    for k in sorted(set(kinds)):
        c = cnt[kinds == k]
        print(f"   kind {k}: {len(c)} tiles, neighbours max {c.max()} mean {c.mean():.1f}")

stats("dag10k", synth.random_dag(10000, 4, 64, 4, seed=1))
stats("mixed-arity 10k", synth.random_dag(10000, 4, 64, [2, 3, 4], seed=8))
stats("grid 64x64", synth.grid(64, 64, 4, seed=2))
stats("grid 128x128", synth.grid(128, 128, 4, seed=2))
stats("dag 1000", synth.random_dag(1000, 4, 32, 4, seed=3))
stats("dag 3000 window 512", synth.random_dag(3000, 4, 512, 4, seed=3))

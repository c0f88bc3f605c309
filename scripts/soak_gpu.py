#!/usr/bin/env python3
"""Randomised parity soak on the GPU box (not part of the test suite; a few minutes):  python scripts/soak_gpu.py [seconds] [seed]
Random networks (grids, chains, DAGs of mixed arity and in-degree) x random evidence x eps; for each: the default path against the oracle
(equal sweep counts; bit-equal marginals where every node has <= 2 parents, <= 1e-12 otherwise), every eligible one-launch path forced,
the DAG path's dataflow form against its barrier form (bit for bit), a batch of three sets against its single runs, and a fresh engine
against a reloaded one.  Prints one line per network and a summary; exits non-zero on the first difference."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (the checker)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
FORCE = {0: {"multisweep": 0}, 2: {"multisweep": 2, "small": 0, "mid": 0, "dag": 0}, 3: {"small": 2, "mid": 0, "dag": 0},
         4: {"mid": 2, "small": 0, "dag": 0}, 5: {"dag": 2}}
DEFAULTS = {"multisweep": 1, "small": 1, "mid": 1, "dag": 1, "dagflow": 0}
ELIGIBLE = {2: "resident_eligible", 3: "small_eligible", 4: "mid_eligible", 5: "dag_eligible"}


def network(i):
    kind = rng.integers(0, 5)
    seed = int(rng.integers(1, 1 << 30))
    if kind == 0:
        r, c = int(rng.integers(2, 90)), int(rng.integers(2, 90))
        return f"grid{r}x{c}", synth.grid(r, c, int(rng.choice([2, 3, 4])), seed=seed)
    if kind == 1:
        n = int(rng.integers(5, 600))
        return f"chain{n}", synth.random_dag(n, 1, int(rng.integers(1, 8)), int(rng.choice([2, 3, 4, 5])), seed=seed)
    n = int(rng.choice([30, 80, 200, 500, 1200, 3000, 6000]))
    mp = int(rng.integers(2, 6))
    arities = [4] if kind == 2 else [int(x) for x in rng.choice([2, 3, 4, 5, 6], size=int(rng.integers(1, 5)))]
    if kind == 3:
        arities = [int(x) for x in rng.choice([2, 3, 4], size=int(rng.integers(1, 4)))]
    return f"dag{n}_p{mp}_k{''.join(map(str, arities))}", synth.random_dag(n, mp, int(rng.choice([8, 32, 64, 256])), arities if len(arities) > 1 else arities[0], seed=seed)


def same(a, b, exact):
    if a["sweeps"] != b["sweeps"]:
        return False
    if exact:
        return np.array_equal(a["beliefs"], b["beliefs"], equal_nan=True)
    nan = np.isnan(b["beliefs"])
    return np.array_equal(np.isnan(a["beliefs"]), nan) and (nan.all() or np.abs(a["beliefs"][~nan] - b["beliefs"][~nan]).max() < 1e-12)


t_end = time.time() + budget
count, checks = 0, 0
while time.time() < t_end:
    name, g = network(count)
    exact = int(np.diff(g.in_ptr).max()) <= 2 if g.n else True
    # (the tile kernels' any-arity variant keeps the reference's order for tables of up to 128 entries; a two-parent node of arity 6 has 216)
    exact_tiles = exact and int(np.diff(g.cpt_off).max()) <= 128
    ev = synth.random_evidence(g, float(rng.choice([0.0, 0.02, 0.1, 0.3])), seed=int(rng.integers(1, 1 << 30)))
    eps = float(rng.choice([1e-3, 1e-6, 1e-9]))
    cap = int(rng.choice([0, 0, 0, 3, 40]))
    want = oracle.bp_run(g, ev, eps, cap)
    paths = []
    with Engine(g) as e:
        def opts(d):
            for k, v in {**DEFAULTS, **d}.items():
                e.set_option(k, v)
        opts({})
        got = e.bp_run(ev, eps, cap)
        default_path = e.last_path()
        assert same(got, want, (exact_tiles if default_path in (0, 2) else exact) or default_path in (3, 4)), (name, "default path", default_path, got["sweeps"], want["sweeps"])
        checks += 1
        for path, force in FORCE.items():
            if path in ELIGIBLE and not e.info(ELIGIBLE[path]):
                continue
            opts(force)
            r = e.bp_run(ev, eps, cap)
            if e.last_path() != path:
                continue
            paths.append(path)
            # the item kernels (3, 4) keep the oracle's order for any parent count; the others re-associate beyond two parents
            assert same(r, want, (exact_tiles if path in (0, 2) else exact) or path in (3, 4)), (name, "path", path, r["sweeps"], want["sweeps"])
            checks += 1
            if path == 5:
                e.set_option("dagflow", 1)
                f = e.bp_run(ev, eps, cap)
                if e.info("last_dag_flow") == 1:
                    assert f["sweeps"] == r["sweeps"] and np.array_equal(f["beliefs"], r["beliefs"], equal_nan=True), (name, "dataflow form")
                    paths.append("5f")
                    checks += 1
                e.set_option("dagflow", 0)
        opts({})
        if g.n <= 3000:
            sets = [ev, synth.random_evidence(g, 0.05, seed=int(rng.integers(1, 1 << 30))), synth.random_evidence(g, 0.2, seed=int(rng.integers(1, 1 << 30)))]
            out = e.bp_run_batch(sets, eps, cap)
            for q, s in enumerate(sets):
                single = e.bp_run(s, eps, cap)
                assert int(out["sweeps"][q]) == single["sweeps"] and np.array_equal(out["beliefs"][q], single["beliefs"], equal_nan=True), (name, "batch set", q)
            checks += 1
    count += 1
    print(f"{count:4d} {name:28s} n={g.n:5d} eps={eps:g} cap={cap:2d} sweeps={want['sweeps']:4d} default={default_path} forced={paths}", flush=True)
print(f"soak ok: {count} networks, {checks} comparisons")

#!/usr/bin/env python3
"""Randomised parity soak on the GPU box (not part of the test suite; a few minutes):  python scripts/soak_gpu.py [seconds] [seed]
Random networks (grids, chains, DAGs of mixed arity and in-degree) x random evidence x eps; for each: the default path against the oracle
(equal sweep counts; bit-equal marginals where every node has <= 2 parents, <= 1e-12 otherwise), every eligible one-launch path forced,
the DAG path's dataflow form against its barrier form (bit for bit), a batch of three sets against its single runs; every fifth network
as 2-5 shard engines on this one device (emulated all-gather, default stripes or a random owner map) against the unsharded run, every
seventh reloaded with new tables against a fresh engine, every eighth through likelihood weighting against the oracle's histogram.  Prints one line per network and a summary; exits non-zero on the first difference."""
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
skip_until = int(os.environ.get("SOAK_SKIP_UNTIL", "0"))
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
    if rng.random() < 0.25 and g.n:   # soft evidence on a few nodes: positive weights, not normalised (reference :68-73 takes the vector as it is)
        from bayesiannetwork_amd import Evidence
        d = {int(v): (0.05 + rng.random(int(g.k[v]))) for v in rng.choice(g.n, size=min(g.n, 3), replace=False)}
        hard = {int(v): int(np.argmax(ev.val[ev.off[j]:ev.off[j + 1]])) for j, v in enumerate(ev.node) if int(v) not in d}
        ev = Evidence.from_dict(g, {**hard, **d})
    eps = float(rng.choice([1e-3, 1e-6, 1e-9]))
    cap = int(rng.choice([0, 0, 0, 3, 40]))
    special = rng.random()
    if special < 0.04 and g.n <= 1200:      # a run beyond one launch's budget of 1 024 iterations (every one-launch path continues from its state in memory)
        eps, cap = 0.0, int(rng.choice([1030, 2060]))
    elif special < 0.08 and g.n:            # an all-zero evidence vector: 0 / 0 -> NaN in the reference (no zero guard, :298-311); the NaNs must coincide
        from bayesiannetwork_amd import Evidence
        v0 = int(rng.integers(0, g.n))
        hard = {int(v): (ev.val[ev.off[j]:ev.off[j + 1]].copy()) for j, v in enumerate(ev.node)}
        hard[v0] = np.zeros(int(g.k[v0]))
        ev = Evidence.from_dict(g, hard)
        cap = cap or 12
    if count + 1 < skip_until and skip_until > 0:   # SOAK_SKIP_UNTIL=N: replay the draws of the first N - 1 networks without running them (to get back to a failure)
        if g.n <= 3000:
            rng.integers(1, 1 << 30); rng.integers(1, 1 << 30)
        if count % 5 == 0 and 4 <= g.n <= 4000:
            nr = int(rng.integers(2, 6))
            if not rng.random() < 0.6:
                rng.integers(0, nr, size=g.n)
            rng.integers(0, 2)
        if count % 7 == 0 and g.n <= 6000:
            rng.integers(1, 1 << 30)
        if count % 8 == 0 and g.n <= 6000:
            rng.integers(1, 1 << 30); rng.integers(1, 1 << 30)
        count += 1
        continue
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
            batch_path = e.last_path()
            for q, s in enumerate(sets):
                single = e.bp_run(s, eps, cap)
                assert int(out["sweeps"][q]) == single["sweeps"] and np.array_equal(out["beliefs"][q], single["beliefs"], equal_nan=True), \
                    (name, "batch set", q, "batch path", batch_path, "single path", e.last_path(), "sweeps", int(out["sweeps"][q]), single["sweeps"],
                     "max difference", float(np.nanmax(np.abs(out["beliefs"][q] - single["beliefs"]))), "eps", eps, "cap", cap)
            checks += 1
    if count % 5 == 0 and 4 <= g.n <= 4000:   # the multi-GPU data path on ONE device: 2-5 shard engines, emulated all-gather, against the unsharded tile kernels
        from bayesiannetwork_amd import engine as engine_mod
        nranks = int(rng.integers(2, 6))
        owner = None if rng.random() < 0.6 else rng.integers(0, nranks, size=g.n).astype(np.int32)   # default stripes, or a random map (cuts almost every edge)
        with Engine(g) as single:
            for k in ("small", "mid", "dag"):
                single.set_option(k, 0)
            ws = single.bp_run(ev, eps, cap)
        shards = [Engine(g, rank=r, nranks=nranks, owner=owner) for r in range(nranks)]
        try:
            out = engine_mod.run_shards_on_one_device(shards, ev, eps, cap, overlapped=bool(rng.integers(0, 2)))
            bel = sum(sh.bp_beliefs() for sh in shards)
        finally:
            for sh in shards:
                sh.close()
        assert out["sweeps"] == ws["sweeps"] and np.array_equal(bel, ws["beliefs"], equal_nan=True), (name, "shards", nranks, owner is not None)
        checks += 1
    if count % 7 == 0 and g.n <= 6000:   # new tables on the same structure: a reloaded engine against a fresh one, on the default path
        from bayesiannetwork_amd import FlatModel
        from bayesiannetwork_amd.synth import _random_cpts
        _, cpt2 = _random_cpts(g.k, g.in_ptr, g.in_idx, int(rng.integers(1, 1 << 30)))
        g2 = FlatModel(g.k, g.in_ptr, g.in_idx, g.cpt_off, cpt2)
        with Engine(g) as a, Engine(g2) as b:
            a.bp_run(ev, eps, cap)
            a.reload_cpt(g2.cpt)
            ra, rb = a.bp_run(ev, eps, cap), b.bp_run(ev, eps, cap)
            assert a.last_path() == b.last_path() and ra["sweeps"] == rb["sweeps"] and np.array_equal(ra["beliefs"], rb["beliefs"], equal_nan=True), (name, "reload")
        checks += 1
    if count % 8 == 0 and g.n <= 6000:   # likelihood weighting on the same network: the weighted histogram of 2 048 samples against the oracle's
        st = synth.random_evidence(g, 0.05, seed=int(rng.integers(1, 1 << 30))).hard_states(g)
        seed = int(rng.integers(1, 1 << 30))
        with Engine(g) as e:
            h = e.lw_run(st, 2048, seed=seed)
        o = oracle.lw_run(g, st, 2048, seed=seed)
        assert np.allclose(h, o["hist"], rtol=1e-9, atol=1e-12), (name, "likelihood weighting", float(np.abs(h - o["hist"]).max()))
        checks += 1
    count += 1
    print(f"{count:4d} {name:28s} n={g.n:5d} eps={eps:g} cap={cap:2d} sweeps={want['sweeps']:4d} default={default_path} forced={paths}", flush=True)
print(f"soak ok: {count} networks, {checks} comparisons")

"""The any-arity tile variant (one wavefront per node, `tile_flat`): mixed arities, arities above 4,
tables above the ordered-path limit, many children, evidence, shards.

Versus the C restatement: bit-identical where the reference's arithmetic order is deterministic
(every node has <= 2 parents AND <= 128 CPT entries: the ordered path), to rounding otherwise (>= 3
parents multiply in unordered_map order in the reference itself, and tables above 128 entries are
accumulated with LDS atomics); the stopping sweep is the same everywhere."""
import numpy as np
import pytest

from bayesiannetwork_amd import Evidence, from_parent_lists, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine(bnlib):
    from bayesiannetwork_amd.engine import Engine
    return Engine


def variants(eng):
    return sorted({c["variant"] for c in eng.layout_classes()})


def star(n_children, k_root, k_child, seed):
    """One root with many children (parent role beyond the staged 256 doubles when n*k > 256)."""
    rng = np.random.default_rng(seed)
    parents = [[]] + [[0] for _ in range(n_children)]
    ks = [k_root] + [k_child] * n_children
    cpts = []
    for v, ps in enumerate(parents):
        rows = int(np.prod([ks[p] for p in ps])) if ps else 1
        t = 0.1 + 0.9 * rng.random((rows, ks[v]))
        cpts.append(t / t.sum(axis=1, keepdims=True))
    return from_parent_lists(ks, parents, cpts)


CASES = [
    # name, model factory, exact (bit-identical to the oracle expected)
    ("mixed_le2", lambda: synth.random_dag(300, 2, 24, [2, 3, 4, 5, 3, 2, 4], seed=31), True),
    ("k5_le2", lambda: synth.random_dag(200, 2, 16, 5, seed=32), True),       # 5*25 = 125 entries: ordered path
    ("k7_le2", lambda: synth.random_dag(120, 2, 16, 7, seed=33), False),      # 343 entries: atomics
    ("mixed_le4", lambda: synth.random_dag(400, 4, 32, [2, 3, 4, 3, 2, 4, 4], seed=34), False),
    ("k3_le5", lambda: synth.random_dag(300, 5, 32, 3, seed=35), False),      # up to 729 entries
    ("k2_le8", lambda: synth.random_dag(300, 8, 32, 2, seed=36), False),      # 8 parents: the variant's limit
    ("k9_le1", lambda: synth.random_dag(150, 1, 16, 9, seed=37), True),
    ("star_small", lambda: star(40, 5, 3, seed=38), True),                   # 40 * 5 = 200 staged doubles
    ("star_big", lambda: star(120, 3, 5, seed=39), True),                    # 360 > 256: memory fallback
]


@pytest.mark.parametrize("name,make,exact", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("frac,eps", [(0.0, 1e-3), (0.1, 1e-7)])
def test_flat_variant_matches_oracle(Engine, oracle_mod, name, make, exact, frac, eps):
    m = make()
    ev = synth.random_evidence(m, frac, seed=5) if frac else Evidence.none()
    want = oracle_mod.bp_run(m, ev, eps=eps, dump_msgs=True)
    with Engine(m) as eng:
        assert 3 in variants(eng), "the case is meant to exercise the flat variant"
        got = eng.bp_run(ev, eps)
        res = eng.bp_residuals()
        pi, lam = eng.bp_messages()
    assert got["sweeps"] == want["sweeps"]
    if exact:
        assert np.array_equal(got["beliefs"], want["beliefs"], equal_nan=True)
        assert np.array_equal(res, want["residuals"])
        assert np.array_equal(pi, want["pi_msg"], equal_nan=True) and np.array_equal(lam, want["lambda_msg"], equal_nan=True)
    else:
        assert np.allclose(got["beliefs"], want["beliefs"], rtol=1e-11, atol=1e-14)
        assert np.allclose(res, want["residuals"], rtol=1e-8, atol=1e-15)
        assert np.allclose(pi, want["pi_msg"], rtol=1e-11, atol=1e-14) and np.allclose(lam, want["lambda_msg"], rtol=1e-11, atol=1e-14)


def many_parents(m, seed):
    """m binary roots, one binary node with all of them as parents (2^m rows), and a child of that node"""
    rng = np.random.default_rng(seed)
    parents = [[] for _ in range(m)] + [list(range(m))] + [[m]]
    ks = [2] * (m + 2)
    cpts = []
    for v, ps in enumerate(parents):
        t = 0.1 + 0.9 * rng.random((2 ** len(ps), 2))
        cpts.append(t / t.sum(axis=1, keepdims=True))
    return from_parent_lists(ks, parents, cpts)


@pytest.mark.parametrize("m", [9, 12, 16])
def test_nodes_with_9_to_16_parents(Engine, oracle_mod, m):
    """The reference's cpt_t / graph_t accept any in-degree (graph.hpp:57-154, :490-525); this engine up to BN_MAX_PARENTS = 16.
    Nodes with 9-16 parents are beyond the lane-group and any-arity variants (<= 8 parents) and run on the one-lane generic
    tile: a 12-parent binary node has a 4 096-row table, a 16-parent one 65 536 rows (1 MB).  Same sweep count as the oracle,
    marginals within 1e-9 (the reference's own products over >= 3 parents are unordered, belief_propagation.hpp:253)."""
    g = many_parents(m, seed=50 + m)
    evs = [Evidence.none(), Evidence.from_dict(g, {0: 1, m - 1: 0, m + 1: 1}), Evidence.from_dict(g, {m: 0})]
    with Engine(g) as eng:
        assert eng.info("small_eligible") == 0 and eng.info("mid_eligible") == 0 and eng.info("dag_eligible") == 0
        for ev in evs:
            want = oracle_mod.bp_run(g, ev, 1e-9, dump_msgs=True)
            got = eng.bp_run(ev, 1e-9)
            assert eng.last_path() == 0 and got["sweeps"] == want["sweeps"]
            assert np.allclose(got["beliefs"], want["beliefs"], rtol=1e-9, atol=1e-14)
            assert np.allclose(eng.bp_residuals(), want["residuals"], rtol=1e-8, atol=1e-15)
            pi, lam = eng.bp_messages()
            assert np.allclose(pi, want["pi_msg"], rtol=1e-9, atol=1e-14) and np.allclose(lam, want["lambda_msg"], rtol=1e-9, atol=1e-14)
        hist = eng.lw_run(evs[1].hard_states(g), 20000, seed=3)     # the samplers read the same tables
        want = oracle_mod.lw_run(g, evs[1].hard_states(g), 20000, seed=3)["hist"]
        assert np.allclose(hist, want, rtol=1e-9, atol=1e-12)


def test_flat_equals_one_lane_generic_path(Engine):
    """lanes_per_node = 1 keeps the old one-lane-per-node generic path: same sweeps, same marginals."""
    m = synth.random_dag(250, 3, 24, [2, 3, 4, 5], seed=41)
    ev = synth.random_evidence(m, 0.05, seed=6)
    with Engine(m) as a, Engine(m, lanes_per_node=1) as b:
        assert 3 in variants(a) and 3 not in variants(b)
        ra, rb = a.bp_run(ev, 1e-6), b.bp_run(ev, 1e-6)
    assert ra["sweeps"] == rb["sweeps"]
    assert np.allclose(ra["beliefs"], rb["beliefs"], rtol=1e-11, atol=1e-14)


def test_flat_tiles_in_shards(Engine):
    """Flat tiles with parents on another rank (in-edge references into the exchange region)."""
    from bayesiannetwork_amd.engine import run_shards_on_one_device
    m = synth.random_dag(240, 3, 40, [2, 3, 5, 4], seed=43)
    ev = synth.random_evidence(m, 0.05, seed=7)
    with Engine(m) as one:
        one.set_option("mid", 0)   # the shards run the tile kernels: the reference run too (same bits)
        want = one.bp_run(ev, 1e-6)
    for nranks in (2, 3):
        owner = (np.arange(m.n) * nranks // m.n).astype(np.int32)
        engines = [Engine(m, rank=r, nranks=nranks, owner=owner) for r in range(nranks)]
        try:
            got = run_shards_on_one_device(engines, ev, 1e-6)
            bel = sum(e.bp_beliefs() for e in engines)  # zeros for nodes of other ranks
        finally:
            for e in engines:
                e.close()
        assert got["sweeps"] == want["sweeps"]
        assert np.array_equal(bel, want["beliefs"], equal_nan=True)


def _random_network(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(5, 60))
    maxp = int(rng.integers(1, 7))
    ks = [int(x) for x in rng.integers(1, 8, size=n)]
    parents, cpts = [], []
    for v in range(n):
        m = int(rng.integers(0, min(maxp, v) + 1))
        ps = sorted(rng.choice(v, size=m, replace=False).tolist()) if m else []
        while ps and ks[v] * int(np.prod([ks[p] for p in ps])) > 20000:
            ps.pop()
        parents.append(ps)
        rows = int(np.prod([ks[p] for p in ps])) if ps else 1
        t = 0.05 + rng.random((rows, ks[v]))
        cpts.append(t / t.sum(axis=1, keepdims=True))
    m = from_parent_lists(ks, parents, cpts)

    def evidence():
        evd = {}
        for v in rng.choice(n, size=max(1, n // 8), replace=False):
            v = int(v)
            evd[v] = int(rng.integers(0, ks[v])) if rng.random() < 0.7 else (0.1 + rng.random(ks[v]))
        return Evidence.from_dict(m, evd)
    return m, parents, cpts, evidence


# BN_STRESS_SEEDS=400 python -m pytest tests/test_flat_gpu.py -m gpu   for a longer one-off run
_SEEDS = int(__import__("os").environ.get("BN_STRESS_SEEDS", "12"))


@pytest.mark.parametrize("seed", range(_SEEDS))
def test_random_shapes_match_oracle(Engine, oracle_mod, seed):
    """Small random networks with arities 1..7 (arity 1 included), 0..6 parents, random hard and
    soft evidence: same stopping sweep, marginals to rounding (bit-identical when no node has more
    than two parents and no table exceeds the ordered path's 128 entries)."""
    m, parents, cpts, evidence = _random_network(seed)
    ev = evidence()
    want = oracle_mod.bp_run(m, ev, eps=1e-6, max_sweeps=200)
    with Engine(m) as eng:
        got = eng.bp_run(ev, 1e-6, max_sweeps=200)
    assert got["sweeps"] == want["sweeps"]
    exact = max(len(p) for p in parents) <= 2 and max(c.size for c in cpts) <= 128
    if exact:
        assert np.array_equal(got["beliefs"], want["beliefs"], equal_nan=True)
    else:
        assert np.allclose(got["beliefs"], want["beliefs"], rtol=1e-10, atol=1e-13, equal_nan=True)


@pytest.mark.parametrize("seed", range(max(_SEEDS // 2, 6)))
def test_random_shapes_batch_equals_single(Engine, seed):
    """The same random networks, three evidence sets per call: every set gets the bits of its single run
    (per-sweep launches with one set per blockIdx.y, on the dense second engine where the layout has one)."""
    m, _, _, evidence = _random_network(seed)
    evs = [evidence(), None, evidence()]
    with Engine(m) as eng:
        want = [eng.bp_run(ev, 1e-6, max_sweeps=200) for ev in evs]
        got = eng.bp_run_batch(evs, 1e-6, max_sweeps=200)
    for q, w in enumerate(want):
        assert got["sweeps"][q] == w["sweeps"]
        assert np.array_equal(got["beliefs"][q], w["beliefs"], equal_nan=True)


def test_eight_parents_on_the_ordered_path(Engine, oracle_mod):
    """Eight parents with a table of <= 128 entries (some parents of arity 1): the ordered path stages the terms of
    pi(v) and of seven lambda-messages in one pass and the eighth parent's in a second one (bn_tiles.hpp)."""
    rng = np.random.default_rng(77)
    ks = [1, 2, 1, 2, 1, 2, 1, 3, 3, 2, 3]            # nodes 0..7: roots, the parents of node 8; 9 and 10: its children
    parents = [[] for _ in range(8)] + [list(range(8)), [8], [3, 8]]
    cpts = []
    for v, ps in enumerate(parents):
        rows = int(np.prod([ks[p] for p in ps])) if ps else 1
        t = 0.1 + rng.random((rows, ks[v]))
        cpts.append(t / t.sum(axis=1, keepdims=True))
    m = from_parent_lists(ks, parents, cpts)
    assert ks[8] * int(np.prod(ks[:8])) <= 128
    for evd in ({}, {9: 1, 1: 0}, {10: np.array([0.2, 0.5, 0.3]), 8: 2}):
        ev = Evidence.from_dict(m, evd)
        want = oracle_mod.bp_run(m, ev, eps=1e-9, max_sweeps=100)
        with Engine(m) as eng:
            assert 3 in variants(eng)
            got = eng.bp_run(ev, 1e-9, max_sweeps=100)
        assert got["sweeps"] == want["sweeps"]
        assert np.allclose(got["beliefs"], want["beliefs"], rtol=1e-12, atol=1e-14)

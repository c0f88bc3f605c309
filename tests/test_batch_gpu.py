"""Several evidence sets per call (bn_bp_run_batch): an extension beside the drop-in -- the reference runs
one query per operator() call (belief_propagation.hpp:31) -- so the bar is that every set of a batch gets
exactly what running it alone gives: same sweep count (sets stop on different sweeps), same residual
history, same bits in the marginals; on the resident path (sets walked round-robin in one launch, 4 at a
time) and on the per-sweep launches with one evidence set per blockIdx.y that every other network takes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _without_the_dag_path(monkeypatch):
    """These tests are about the tile kernels' batches; k = 4 networks that fit the chip would by default take the register-resident DAG
    path (bn_dag.hip, tests/test_dag_gpu.py).  BN_DAG sets the option's default for engines created from here on."""
    monkeypatch.setenv("BN_DAG", "0")


@pytest.fixture(scope="module")
def Engine(bnlib):
    from bayesiannetwork_amd.engine import Engine
    return Engine


def _alone(eng, ev, eps, max_sweeps=0):
    eng.set_option("multisweep", 0)
    r = eng.bp_run(ev, eps, max_sweeps)
    return r, eng.bp_residuals()


def _check_batch(eng, evs, eps, max_sweeps=0, want_path=None, reps=2):
    want = [_alone(eng, ev, eps, max_sweeps) for ev in evs]
    eng.set_option("multisweep", 2)
    for _ in range(reps):
        out = eng.bp_run_batch(evs, eps, max_sweeps)
        if want_path is not None:
            assert eng.last_path() == want_path
        for q, (r, hist) in enumerate(want):
            assert out["sweeps"][q] == r["sweeps"], f"set {q}"
            assert np.array_equal(out["beliefs"][q], r["beliefs"], equal_nan=True), f"set {q}"
            assert out["residual"][q] == r["residual"] or (np.isnan(out["residual"][q]) and np.isnan(r["residual"]))
            assert np.array_equal(eng.bp_residuals_batch(q), hist), f"set {q}"
    return [r["sweeps"] for r, _ in want]


@pytest.mark.parametrize("rows,cols,k,n_sets,eps", [(316, 316, 4, 4, 1e-3), (316, 316, 4, 8, 1e-3), (64, 64, 4, 3, 1e-6),
                                                     (40, 33, 3, 5, 1e-6), (7, 5, 4, 8, 1e-9), (50, 50, 2, 2, 1e-6)])
def test_batch_equals_single_runs_grids(Engine, rows, cols, k, n_sets, eps):
    from bayesiannetwork_amd import synth
    g = synth.grid(rows, cols, k, seed=rows * 7 + cols)
    # different amounts of evidence -> the sets converge on different sweeps
    evs = [synth.random_evidence(g, f, seed=11 + q) for q, f in enumerate([0.0, 0.01, 0.05, 0.2, 0.002, 0.1, 0.0, 0.03][:n_sets])]
    with Engine(g) as eng:
        eng.set_option("small", 0)   # (the 7 x 5 grid would otherwise take the one-workgroup path: tests/test_small_gpu.py)
        sweeps = _check_batch(eng, evs, eps, want_path=2)
        if rows >= 40:
            assert len(set(sweeps)) > 1, "the case should exercise sets leaving the rotation at different sweeps"


def test_batch_caps_and_reuse(Engine):
    from bayesiannetwork_amd import Evidence, synth
    g = synth.grid(48, 48, 4, seed=5)
    evs = [synth.random_evidence(g, 0.03, seed=2), None, Evidence.from_dict(g, {5: np.array([0.2, 0.5, 0.2, 0.1]), 900: 2})]
    with Engine(g) as eng:
        eng.set_option("mid", 0)   # the resident kernel's batch form (by default: tests/test_mid_gpu.py)
        _check_batch(eng, evs, 1e-12, max_sweeps=3, want_path=2)   # every set capped together
        _check_batch(eng, evs, 1e-3, want_path=2)
        _check_batch(eng, evs[:1], 1e-6, want_path=2)              # a batch of one
        _check_batch(eng, list(reversed(evs)) + evs, 1e-6, want_path=2)   # larger batch on the same engine
        r, _ = _alone(eng, evs[0], 1e-6)                           # single-query calls still work in between
        eng.set_option("multisweep", 1)
        assert np.array_equal(eng.bp_run(evs[0], 1e-6)["beliefs"], r["beliefs"])


def test_batch_on_per_sweep_launches_every_tile_variant(Engine):
    """Networks the resident kernel does not cover: every per-sweep launch carries all sets (blockIdx.y)."""
    import os
    from bayesiannetwork_amd import _lib, synth
    from bayesiannetwork_amd.dsc import load_dsc
    d = synth.random_dag(3000, 4, 64, 4, seed=5)                    # register-resident + lane-group tiles
    evs = [synth.random_evidence(d, f, seed=q) for q, f in enumerate([0.0, 0.02, 0.1])]
    with Engine(d) as eng:
        eng.set_option("dag", 0)   # (by default this network takes the register-resident DAG path: tests/test_dag_gpu.py)
        sweeps = _check_batch(eng, evs, 1e-6, want_path=0)
        assert len(set(sweeps)) > 1
        _check_batch(eng, evs, 1e-12, max_sweeps=4, want_path=0)    # every set capped together
        _check_batch(eng, evs + evs[::-1] + evs + evs, 1e-4, want_path=0)   # 12 sets, larger batch on the same engine
        _check_batch(eng, evs[:1], 1e-9, want_path=0)               # a batch of one, more sweeps than predicted
    t = synth.random_dag(2000, 2, 8, 4, seed=41)                    # <= 2 parents: one-lane tiles, resident if <= 8 children
    evs = [synth.random_evidence(t, f, seed=q) for q, f in enumerate([0.0, 0.02, 0.1])]
    with Engine(t) as eng:
        _check_batch(eng, evs, 1e-6)
    m = synth.random_dag(400, 3, 24, [2, 3, 4], seed=9)            # any-arity tiles (+ register-resident ones)
    evs = [synth.random_evidence(m, f, seed=q) for q, f in enumerate([0.0, 0.05, 0.1, 0.02])]
    with Engine(m) as eng:
        eng.set_option("mid", 0)   # (by default this network takes the item kernel over several workgroups: tests/test_mid_gpu.py)
        _check_batch(eng, evs, 1e-6, want_path=0)
        with pytest.raises(_lib.BnError):
            eng.bp_run_batch([None] * (_lib.BN_MAX_BATCH_SETS + 1), 1e-3)   # more than BN_MAX_BATCH_SETS
    alarm, _ = load_dsc(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "alarm_shaped.dsc"))
    evs = [synth.random_evidence(alarm, f, seed=q) for q, f in enumerate([0.0, 0.05, 0.1, 0.2] * 4)]
    with Engine(alarm) as eng:                                      # any-arity tiles only: the light kernel, 16 sets
        eng.set_option("small", 0)
        _check_batch(eng, evs, 1e-9, want_path=0)
        _check_batch(eng, evs[:5], 1e-3, max_sweeps=2, want_path=0)
        eng.set_option("small", 1)                                   # by default: one workgroup per set (bn_small.hip), same bits
        _check_batch(eng, evs, 1e-9, want_path=3)


def test_batch_more_sets_than_one_resident_launch_walks(Engine):
    from bayesiannetwork_amd import synth
    g = synth.grid(40, 40, 4, seed=3)
    evs = [synth.random_evidence(g, 0.01 * (q % 5), seed=q) for q in range(19)]   # 5 launches: 4 + 4 + 4 + 4 + 3 sets (kResidentMaxSets = 4, balanced chunks)
    with Engine(g) as eng:
        eng.set_option("mid", 0)
        _check_batch(eng, evs, 1e-6, want_path=2, reps=1)
        eng.set_option("multisweep", 0)                              # the same batch through the per-sweep launches
        out = eng.bp_run_batch(evs, 1e-6)
        assert eng.last_path() == 0
        for q, ev in enumerate(evs):
            r = eng.bp_run(ev, 1e-6)
            assert out["sweeps"][q] == r["sweeps"] and np.array_equal(out["beliefs"][q], r["beliefs"])


def test_batch_resident_runs_beyond_one_launch(Engine):
    """Runs of more than a launch's 1 024 iterations, in a batch of two chunks on the resident path: all chunks' first launches are
    enqueued behind each other (one host wait), the sets that are not finished go on chunk by chunk -- sweep counts, bits and residual
    histories of the single runs (eps = 0: every set is capped at 1 030 sweeps)."""
    from bayesiannetwork_amd import synth
    g = synth.grid(24, 24, 4, seed=8)
    evs = [synth.random_evidence(g, 0.02 * (q % 3), seed=40 + q) for q in range(6)]
    with Engine(g) as eng:
        eng.set_option("mid", 0)
        eng.set_option("small", 0)
        _check_batch(eng, evs, 0.0, max_sweeps=1030, want_path=2, reps=1)
        assert eng.bp_stats()["sweep_launches"] == 4    # two chunks, two launches each
        _check_batch(eng, evs, 1e-9, want_path=2, reps=1)   # (and an ordinary batch on the same engine afterwards)


def test_mirror_class_run_batch(bnlib):
    """BeliefPropagation.run_batch (the Python spelling of the drop-in's run_batch extension): entry q == bp(queries[q])."""
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import BeliefPropagation
    m = synth.resume_chain()
    bp = BeliefPropagation(m)
    queries = [{1: np.array([0.0, 0.0, 1.0]), 3: np.array([1.0, 0.0, 0.0])}, {2: np.array([0.0, 1.0])}, {}, {0: np.array([1.0, 0.0, 0.0])}]
    got = bp.run_batch(queries, 1e-9)
    assert len(got) == len(queries)
    for q, pre in enumerate(queries):
        want = bp(pre, 1e-9)
        for v in range(m.n):
            assert np.array_equal(got[q][v], want[v])


def test_batch_beyond_64_sets_on_every_path(Engine):
    """BN_MAX_BATCH_SETS = 256: more than 64 sets per call on the per-sweep launches (blockIdx.y = set), on the resident tiles
    (4 sets per launch, launch after launch) and -- tests/test_small_gpu.py -- one workgroup per set."""
    from bayesiannetwork_amd import synth
    m = synth.random_dag(400, 3, 24, [2, 3, 4], seed=9)
    evs = [synth.random_evidence(m, 0.01 * (q % 7), seed=q) for q in range(130)]
    with Engine(m) as eng:
        eng.set_option("mid", 0)
        _check_batch(eng, evs, 1e-6, want_path=0, reps=1)
    g = synth.grid(40, 40, 4, seed=3)
    evs = [synth.random_evidence(g, 0.01 * (q % 5), seed=q) for q in range(70)]
    with Engine(g) as eng:
        eng.set_option("mid", 0)
        _check_batch(eng, evs, 1e-4, want_path=2, reps=1)


def test_batch_keeps_single_query_bits_where_the_dense_layout_would_not(bnlib, oracle_mod):
    """"Each set gets exactly the result its single query gives it" (bn_mi355x.h).  Batches of a network whose layout was built for the
    latency of one query normally run on a second engine with the dense layout; where the two layouts put a node with a table of more
    than 128 entries on different tile variants (they sum in different orders) the batch stays on the engine's own layout instead
    (scripts/soak_gpu.py found a 200-node network of arities {4, 6} whose batch differed from its single queries by 1e-16; round 6)."""
    import numpy as np
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import Engine
    refused = 0
    for seed in range(1, 9):
        g = synth.random_dag(200, 4, 32, [4, 6, 4], seed=seed)
        sets = [synth.random_evidence(g, f, seed=10 * seed + q) for q, f in enumerate((0.02, 0.05, 0.2))]
        with Engine(g) as eng:
            out = eng.bp_run_batch(sets, 1e-9, 3)
            refused += eng.info("batch_dense_refused")
            for q, ev in enumerate(sets):
                r = eng.bp_run(ev, 1e-9, 3)
                o = oracle_mod.bp_run(g, ev, 1e-9, 3)
                assert r["sweeps"] == o["sweeps"] == int(out["sweeps"][q])
                assert np.array_equal(out["beliefs"][q], r["beliefs"]), (seed, q, float(np.abs(out["beliefs"][q] - r["beliefs"]).max()))
                assert np.abs(r["beliefs"] - o["beliefs"]).max() < 1e-12
    assert refused >= 1   # (at least one of these networks is of the kind the rule is for)

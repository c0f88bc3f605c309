"""The one-launch path (bn_resident.hip: tiles resident in registers / LDS + a grid barrier per sweep):
it must be bit-identical to the per-sweep launch path -- beliefs,
sweep count, per-sweep residuals, final messages -- on every run, also when runs are repeated back to
back (stale cache lines of an earlier run are the classic failure of an in-launch exchange), with and
without evidence, and it must be chosen only for eligible models."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _without_the_dag_path(monkeypatch):
    """These tests are about the resident tiles; k = 4 networks that fit the chip would by default take the register-resident DAG
    path (bn_dag.hip, tests/test_dag_gpu.py).  BN_DAG sets the option's default for engines created from here on."""
    monkeypatch.setenv("BN_DAG", "0")


@pytest.fixture(scope="module")
def Engine(bnlib):
    from bayesiannetwork_amd.engine import Engine
    return Engine


def _launch_path(eng, ev, eps, max_sweeps=0):
    eng.set_option("multisweep", 0)
    r = eng.bp_run(ev, eps, max_sweeps)
    assert eng.last_path() == 0
    return r, eng.bp_residuals(), eng.bp_messages()


def _check_same(eng, ev, eps, want, reps, max_sweeps=0, want_path=None):
    """Both forms of the resident kernel: "flow" 1 = dataflow (a tile waits for its neighbour tiles, the stop decision
    lags an iteration: the state the run ends in must be the one BEFORE the speculative iteration), 0 = grid barrier
    per sweep.  One-block grids run the same LDS-only code under either setting."""
    r0, res0, (pi0, lam0) = want
    eng.set_option("small", 0)   # (the smallest networks here would otherwise take the one-workgroup path: tests/test_small_gpu.py,
    eng.set_option("mid", 0)     #  mid-size ones of k = 4 with two parents the several-workgroup path: tests/test_mid_gpu.py)
    eng.set_option("multisweep", 2)
    # (flow, direct): the barrier form with its two collectors -- every tile block reads all granules itself (default) / the service block
    for flow, direct in ((1, 1), (0, 1), (0, 0)):
        eng.set_option("flow", flow)
        eng.set_option("direct", direct)
        for i in range(reps if flow or direct else max(2, reps // 4)):
            r = eng.bp_run(ev, eps, max_sweeps)
            assert eng.last_path() != 0
            if want_path is not None:
                assert eng.last_path() == want_path
            assert eng.info("last_flow") == (flow if eng.info("flow_eligible") else 0)
            assert r["sweeps"] == r0["sweeps"], f"flow {flow} direct {direct} run {i}"
            assert np.array_equal(r["beliefs"], r0["beliefs"], equal_nan=True), f"flow {flow} direct {direct} run {i}"
            assert r["residual"] == r0["residual"] or (np.isnan(r["residual"]) and np.isnan(r0["residual"]))
        assert np.array_equal(eng.bp_residuals(), res0), f"flow {flow} direct {direct}"
        pi, lam = eng.bp_messages()
        assert np.array_equal(pi, pi0, equal_nan=True) and np.array_equal(lam, lam0, equal_nan=True), f"flow {flow} direct {direct}"
        assert eng.bp_stats()["resident_aborts"] == 0
    eng.set_option("flow", 1)
    eng.set_option("direct", 1)


@pytest.mark.parametrize("rows,cols,k,frac,eps,reps", [
    (316, 316, 4, 0.01, 1e-3, 40), (316, 316, 4, 0.0, 1e-6, 10), (64, 64, 4, 0.05, 1e-9, 10), (40, 33, 3, 0.02, 1e-6, 10),
    (50, 50, 2, 0.02, 1e-6, 10), (1, 300, 4, 0.0, 1e-6, 10), (7, 5, 4, 0.1, 1e-3, 10), (128, 128, 4, 0.01, 1e-6, 10),
    (338, 338, 4, 0.01, 1e-3, 4),   # 1 786 tiles = 224 blocks: the most the path admits (the granule sweep's fourth round is partly filled)
])
def test_resident_equals_launch_path_grids(Engine, rows, cols, k, frac, eps, reps):
    from bayesiannetwork_amd import synth
    g = synth.grid(rows, cols, k, seed=rows * 31 + cols)
    ev = synth.random_evidence(g, frac, seed=3)
    with Engine(g) as eng:
        # every multi-block grid here must really take the dataflow form (a first-column tile of a wide grid has
        # more than 64 neighbour tiles: polled in rounds of 64)
        assert eng.info("flow_eligible") == (1 if eng.info("resident_blocks") > 1 else 0)
        if rows == 338:
            assert eng.info("resident_blocks") == 224
        want = _launch_path(eng, ev, eps)
        _check_same(eng, ev, eps, want, reps, want_path=2)
        # alternate the paths and the evidence: nothing of one run may leak into the next
        ev2 = synth.random_evidence(g, max(frac, 0.02), seed=11)
        want2 = _launch_path(eng, ev2, eps)
        _check_same(eng, ev2, eps, want2, 3, want_path=2)
        _check_same(eng, ev, eps, _launch_path(eng, ev, eps), 3, want_path=2)


def test_resident_trees_and_dags(Engine):
    """<= 2 parents, up to 8 children per node: the all-shapes instantiation of the resident kernel (4- and
    8-children parent roles).  The dense layout (lanes_per_node = 2) keeps nodes with more than 4 children on
    one-lane tiles -- the automatic layout would move them to any-arity tiles, which are never resident."""
    from bayesiannetwork_amd import synth
    resident, many_children = 0, 0
    for seed in range(6):
        d = synth.random_dag(700, 2, 8, [4, 3, 2][seed % 3], seed=40 + seed)
        ev = synth.random_evidence(d, 0.02, seed=seed)
        with Engine(d, lanes_per_node=2) as eng:
            eng.set_option("mid", 0)   # this test is about the resident kernel
            want = _launch_path(eng, ev, 1e-6)
            eng.set_option("multisweep", 2)
            eng.bp_run(ev, 1e-6)
            if eng.last_path() == 0:  # some node has more than 8 children: not eligible, nothing to compare
                continue
            resident += 1
            many_children += int(np.bincount(d.in_idx, minlength=d.n).max() > 4)
            _check_same(eng, ev, 1e-6, want, 5)
    assert resident >= 3 and many_children >= 1


def test_resident_max_sweeps_and_soft_evidence(Engine):
    from bayesiannetwork_amd import Evidence, synth
    g = synth.grid(48, 48, 4, seed=5)
    ev = synth.random_evidence(g, 0.03, seed=2)
    with Engine(g) as eng:
        for cap in (1, 2, 5):
            want = _launch_path(eng, ev, 1e-12, max_sweeps=cap)
            assert want[0]["sweeps"] == cap
            _check_same(eng, ev, 1e-12, want, 2, max_sweeps=cap, want_path=2)
        soft = Evidence.from_dict(g, {5: np.array([0.2, 0.5, 0.2, 0.1]), 700: np.array([0.0, 0.0, 1.0, 0.0]),
                                      1500: np.array([0.25, 0.25, 0.25, 0.25])})
        _check_same(eng, soft, 1e-9, _launch_path(eng, soft, 1e-9), 3, want_path=2)
        # a zero row: 0/0 -> NaN in the reference (no zero guard, belief_propagation.hpp:298-311)
        zero = Evidence.from_dict(g, {100: np.array([0.0, 0.0, 0.0, 0.0])})
        _check_same(eng, zero, 1e-6, _launch_path(eng, zero, 1e-6, max_sweeps=6), 2, max_sweeps=6, want_path=2)


def test_four_waves_per_block_are_chosen_on_the_rounded_block_count(Engine, oracle_mod):
    """A 239 x 240 grid has 898 tiles: 225 blocks of four waves, 232 once rounded up to a multiple of 8 -- more than 0.9 x 256 CUs
    hold beside the barrier's service block.  The engine must see that BEFORE it settles on four waves per block and keep eight
    (120 blocks) instead of losing the resident path for the whole network."""
    from bayesiannetwork_amd import synth
    g = synth.grid(239, 240, 4, seed=6)
    ev = synth.random_evidence(g, 0.01, seed=3)
    o = oracle_mod.bp_run(g, ev, 1e-3, threads=8)
    with Engine(g) as eng:
        assert 897 <= eng.layout()["n_tiles"] <= 916
        assert eng.info("resident_eligible") == 1 and eng.info("resident_waves") == 8
        r = eng.bp_run(ev, 1e-3)
        assert eng.last_path() == 2 and r["sweeps"] == o["sweeps"] and np.array_equal(r["beliefs"], o["beliefs"])
    with Engine(synth.grid(236, 236, 4, seed=6)) as eng:    # 871 tiles: 218 -> 224 blocks of four still fit
        assert eng.info("resident_eligible") == 1 and eng.info("resident_waves") == 4


def test_paths_are_chosen_by_eligibility(Engine):
    from bayesiannetwork_amd import synth
    with Engine(synth.random_dag(300, 3, 32, [2, 3, 4], seed=1)) as eng:  # any-arity tiles: never resident ...
        eng.set_option("multisweep", 2)
        eng.bp_run(None, 1e-3)
        assert eng.last_path() == 4                                         # ... the item kernel over several workgroups instead
        eng.set_option("mid", 0)
        eng.bp_run(None, 1e-3)
        assert eng.last_path() == 0
    with Engine(synth.random_dag(3000, 4, 64, 4, seed=5)) as eng:  # lane-group tiles (3-4 parents): never on the resident TILES --
        eng.set_option("multisweep", 2)
        eng.bp_run(None, 1e-3)
        assert eng.last_path() == 0
        eng.set_option("dag", 1)
        eng.bp_run(None, 1e-3)
        assert eng.last_path() == 5                                 # the register-resident DAG path takes such networks (bn_dag.hip)
    with Engine(synth.grid(64, 64, 4, seed=1)) as eng:  # 66 tiles: four waves per block (one per SIMD), where the resident
        eng.bp_run(None, 1e-3)                           # kernel beats the launches from the smallest networks on
        assert eng.last_path() == 2 and eng.info("resident_waves") == 4
    with Engine(synth.grid(250, 250, 4, seed=1)) as eng:  # 978 tiles: too many for four waves per block
        assert eng.info("resident_waves") == 8
    with Engine(synth.grid(20, 20, 4, seed=1)) as eng:
        eng.set_option("multisweep", 2)
        eng.bp_run(None, 1e-3)
        assert eng.last_path() == 4   # k = 4, two parents, 400 nodes: the item kernel over several workgroups is faster than the tiles
        eng.set_option("mid", 0)
        eng.bp_run(None, 1e-3)
        assert eng.last_path() == 2
        eng.set_option("multisweep", 0)
        eng.bp_run(None, 1e-3)
        assert eng.last_path() == 0


@pytest.mark.parametrize("flow", [1, 0])
def test_resident_run_longer_than_one_launch(Engine, flow):
    """A launch executes at most 1024 iterations; a longer run continues in another launch from the state the first
    one left in memory (dataflow form: the node vectors every iteration stores, the record buffer of the right parity)."""
    from bayesiannetwork_amd import synth
    g = synth.grid(40, 33, 3, seed=77)   # 21 tiles in 3 blocks
    ev = synth.random_evidence(g, 0.02, seed=5)
    with Engine(g) as eng:
        want = _launch_path(eng, ev, 0.0, max_sweeps=1100)   # eps = 0: never converges, stopped at max_sweeps
        assert want[0]["sweeps"] == 1100
        eng.set_option("multisweep", 2)
        eng.set_option("flow", flow)
        for _ in range(2):
            r = eng.bp_run(ev, 0.0, 1100)
            assert eng.last_path() == 2 and eng.bp_stats()["sweep_launches"] == 2
            assert r["sweeps"] == 1100
            assert np.array_equal(r["beliefs"], want[0]["beliefs"])
            assert np.array_equal(eng.bp_residuals(), want[1])
        pi, lam = eng.bp_messages()
        assert np.array_equal(pi, want[2][0]) and np.array_equal(lam, want[2][1])
        r = eng.bp_run(ev, 0.0, 1024)                        # exactly one launch's budget
        assert r["sweeps"] == 1024 and np.array_equal(r["beliefs"], _launch_path(eng, ev, 0.0, max_sweeps=1024)[0]["beliefs"])

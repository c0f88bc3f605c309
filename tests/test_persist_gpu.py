"""The persistent dataflow kernel (bn_persist.hip): one launch for the whole run, CPTs held in
registers, tiles synchronised through neighbour flags.  It must be bit-identical to the per-sweep
launch path, pick itself only for eligible models, and survive repeated / interleaved runs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine(bnlib):
    from bayesiannetwork_amd.engine import Engine
    return Engine


def _both(Engine, model, ev, eps, max_sweeps=0):
    with Engine(model) as eng:
        eng.set_option("persistent", 1)
        a = eng.bp_run(ev, eps, max_sweeps)
        pa, ra, (pia, lama) = eng.last_path(), eng.bp_residuals(), eng.bp_messages()
        eng.set_option("persistent", 0)
        b = eng.bp_run(ev, eps, max_sweeps)
        pb, rb, (pib, lamb) = eng.last_path(), eng.bp_residuals(), eng.bp_messages()
    assert pb == 0
    assert a["sweeps"] == b["sweeps"] and a["residual"] == b["residual"]
    assert np.array_equal(a["beliefs"], b["beliefs"], equal_nan=True)
    assert np.array_equal(ra, rb)
    assert np.array_equal(pia, pib, equal_nan=True) and np.array_equal(lama, lamb, equal_nan=True)
    return pa, a


@pytest.mark.parametrize("rows,cols,k,frac,eps", [
    (316, 316, 4, 0.01, 1e-3), (316, 316, 4, 0.0, 1e-6), (64, 64, 4, 0.05, 1e-9), (40, 33, 3, 0.02, 1e-6),
    (50, 50, 2, 0.02, 1e-6), (1, 300, 4, 0.0, 1e-6), (7, 5, 4, 0.1, 1e-3),
])
def test_persistent_equals_launch_path_grids(Engine, rows, cols, k, frac, eps):
    from bayesiannetwork_amd import synth
    g = synth.grid(rows, cols, k, seed=rows * 31 + cols)
    path, _ = _both(Engine, g, synth.random_evidence(g, frac, seed=3), eps)
    assert path == 1, "grids with <= 2 parents and children are eligible"


def test_persistent_many_repeats_config3(Engine, oracle_mod):
    """Hand-off bugs show up as rare wrong values: repeat the full-size run and compare every time."""
    from bayesiannetwork_amd import synth
    g = synth.grid(316, 316, 4, seed=2)
    ev = synth.random_evidence(g, 0.01, seed=7)
    want = oracle_mod.bp_run(g, ev, 1e-3, threads=8)
    with Engine(g) as eng:
        eng.set_option("persistent", 1)   # opt-in: bit-identical but measured slower
        eng.bp_set_evidence(ev)
        for _ in range(40):
            r = eng.bp_run_device(1e-3)
            assert eng.last_path() == 1 and r["sweeps"] == want["sweeps"]
            assert np.array_equal(eng.bp_beliefs(), want["beliefs"])


def test_persistent_max_sweeps_and_eligibility(Engine):
    from bayesiannetwork_amd import synth
    g = synth.grid(30, 30, 4, seed=4)
    path, a = _both(Engine, g, None, 1e-12, max_sweeps=7)
    assert path == 1 and a["sweeps"] == 7
    d = synth.random_dag(400, 4, 32, 4, seed=9)  # 3- and 4-parent nodes use lane groups: not eligible
    with Engine(d) as eng:
        eng.set_option("persistent", 1)
        eng.bp_run(None, 1e-3)
        assert eng.last_path() == 0
    with Engine(g) as eng:
        eng.bp_run(None, 1e-3)
        assert eng.last_path() == 0, "the persistent path is opt-in"
        eng.set_option("persistent", 1)
        eng.bp_run(None, 1e-3)
        assert eng.last_path() == 1

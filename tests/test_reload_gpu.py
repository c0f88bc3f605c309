"""bn_reload_cpt: new CPT values on an unchanged structure.  The reference reads node->cpt at every call
(belief_propagation.hpp:61,186,252; likelihood_weighting.hpp:148-158), so an edited or re-fitted table is seen by the next call;
an engine holds device images made at bn_create -- the lane-striped tile image, the entry tables of the item kernels, the
register image of the DAG path, the sampler's flat copy -- and this call re-derives every one of them.  After it each execution
path answers for the NEW tables exactly as an engine built from them does."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine(bnlib):
    from bayesiannetwork_amd.engine import Engine
    return Engine


def _with_other_cpts(model, seed):
    from bayesiannetwork_amd import FlatModel
    from bayesiannetwork_amd.synth import _random_cpts
    _, cpt = _random_cpts(model.k, model.in_ptr, model.in_idx, seed)
    return FlatModel(model.k, model.in_ptr, model.in_idx, model.cpt_off, cpt, name=model.name + "_other")


CASES = [("grid48_resident_and_launches", lambda s: s.grid(48, 48, 4, seed=5), [{"multisweep": 2, "mid": 0}, {"multisweep": 0}]),
         ("alarm_sized_one_workgroup", lambda s: s.random_dag(30, 4, 12, [2, 3, 4, 3], seed=7), [{}]),
         ("mixed300_several_workgroups", lambda s: s.random_dag(300, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=12), [{}, {"mid": 0}]),
         ("dag800_register_resident", lambda s: s.random_dag(800, 4, 48, 4, seed=41), [{}, {"dag": 0}]),
         ("root_heavy", lambda s: s.random_dag(200, 1, 4, 4, seed=9), [{"dag": 2}, {"dag": 0}])]   # roots: their initial pi IS a CPT row


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_reload_equals_a_fresh_engine(Engine, name):
    from bayesiannetwork_amd import synth
    _, make, option_sets = next(c for c in CASES if c[0] == name)
    a = make(synth)
    b = _with_other_cpts(a, seed=99)
    ev = synth.random_evidence(a, 0.05, seed=3)
    for opts in option_sets:
        with Engine(a) as eng, Engine(b) as fresh:
            for k, v in opts.items():
                eng.set_option(k, v)
                fresh.set_option(k, v)
            before = eng.bp_run(ev, 1e-6)
            path = eng.last_path()
            eng.reload_cpt(b.cpt)
            after = eng.bp_run(ev, 1e-6)
            want = fresh.bp_run(ev, 1e-6)
            assert eng.last_path() == path == fresh.last_path()
            assert after["sweeps"] == want["sweeps"] and np.array_equal(after["beliefs"], want["beliefs"]), (name, opts)
            assert np.abs(after["beliefs"] - before["beliefs"]).max() > 1e-3
            eng.reload_cpt(a.cpt)                        # and back
            again = eng.bp_run(ev, 1e-6)
            assert again["sweeps"] == before["sweeps"] and np.array_equal(again["beliefs"], before["beliefs"])
            if a.n <= 1000:                              # batches (where the engine keeps a second, dense copy it is dropped and rebuilt)
                eng.reload_cpt(b.cpt)
                out = eng.bp_run_batch([ev, synth.random_evidence(a, 0.1, seed=4)], 1e-6)
                assert np.abs(out["beliefs"][0] - want["beliefs"]).max() < 1e-12


def test_reload_reaches_the_samplers(Engine, oracle_mod):
    from bayesiannetwork_amd import synth
    a = synth.random_dag(500, 4, 32, 4, seed=3)
    b = _with_other_cpts(a, seed=5)
    ev = synth.random_evidence(a, 0.02, seed=2).hard_states(a)
    with Engine(a) as eng:
        h_a = eng.lw_run(ev, 4096, seed=11)
        eng.reload_cpt(b.cpt)
        h_b = eng.lw_run(ev, 4096, seed=11)
        want = oracle_mod.lw_run(b, ev, 4096, seed=11)["hist"]
        assert np.allclose(h_b, want, rtol=1e-9, atol=1e-12) and not np.allclose(h_a, h_b, rtol=1e-3, atol=1e-6)


def test_reload_argument_errors(Engine):
    from bayesiannetwork_amd import _lib, synth
    a = synth.random_dag(50, 3, 8, 3, seed=1)
    with Engine(a) as eng:
        with pytest.raises(_lib.BnError, match="entries given"):
            eng.reload_cpt(a.cpt[:-1])


def test_reload_before_the_dag_path_is_first_used(Engine):
    """bn_create keeps the register-resident DAG path LIGHT where another path is the default (a 180 x 180 grid: beyond one tile per wave,
    the resident tiles keep it): no padded image on the host, nothing of it on the device.  A reload in that state, then the path's first
    use ("dag" 2), must run the NEW tables; so must "autotune", which tries every eligible path."""
    from bayesiannetwork_amd import synth
    a = synth.grid(180, 180, 4, seed=5)
    b = _with_other_cpts(a, seed=77)
    ev = synth.random_evidence(a, 0.05, seed=3)
    with Engine(a) as eng, Engine(b) as fresh:
        first = eng.bp_run(ev, 1e-6)
        assert eng.last_path() != 5 and eng.info("dag_eligible") == 1
        fresh.set_option("dag", 2)
        want = fresh.bp_run(ev, 1e-6)
        assert fresh.last_path() == 5
        eng.reload_cpt(b.cpt)                    # (the DAG plan is still light here)
        eng.set_option("dag", 2)
        got = eng.bp_run(ev, 1e-6)               # first use: image filled from the NEW tables, uploaded
        assert eng.last_path() == 5 and got["sweeps"] == want["sweeps"] and np.array_equal(got["beliefs"], want["beliefs"])
        assert np.abs(got["beliefs"] - first["beliefs"]).max() > 1e-3
        eng.reload_cpt(a.cpt)                    # (now the full plan is re-derived and re-uploaded)
        back = eng.bp_run(ev, 1e-6)
        assert eng.last_path() == 5 and back["sweeps"] == first["sweeps"] and np.array_equal(back["beliefs"], first["beliefs"])
    with Engine(a) as eng:                       # autotune on an engine whose DAG path was never set up
        eng.set_option("autotune", 1)
        r = eng.bp_run(ev, 1e-6)
        assert eng.info("autotuned") == 1 and r["sweeps"] == first["sweeps"] and np.array_equal(r["beliefs"], first["beliefs"])

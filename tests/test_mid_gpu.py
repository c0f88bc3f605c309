"""Mid-size networks (beyond one workgroup's LDS, not covered by the resident tiles): the items of bn_small.hip spread over
several workgroups by node ranges, state in device memory, a grid barrier per iteration, one launch per run (csrc/bn_mid.hip,
bn_bp_last_path == 4).  Same arithmetic order as the one-workgroup kernel: the oracle BIT FOR BIT, any parent count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine(bnlib):
    from bayesiannetwork_amd.engine import Engine
    return Engine


def _nets():
    from bayesiannetwork_amd import synth
    return [("mixed80", synth.random_dag(80, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=10)),
            ("mixed300", synth.random_dag(300, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=12)),
            ("mixed1000", synth.random_dag(1000, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=14)),     # 25 workgroups
            ("dag60k4_4parents", synth.random_dag(60, 4, 16, 4, seed=5)),                       # 1 024-entry tables: 256-term runs
            ("binary1000_3parents", synth.random_dag(1000, 3, 16, 2, seed=5)),
            ("k7", synth.random_dag(120, 2, 8, [7, 5, 6, 2], seed=8)),
            ("mixed2k_4parents", synth.random_dag(2000, 4, 64, [2, 3, 4, 3, 2, 4, 4], seed=9)),       # 176 k entries: ~150 workgroups
            ("dag600k4_4parents", synth.random_dag(600, 4, 48, 4, seed=78)),                          # parts of the first size do not fit a workgroup: the planner falls back to smaller ones
            ("mixed8000", synth.random_dag(8000, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=19))]             # 318 k entries: 224 workgroups, the most a run may have


@pytest.mark.parametrize("name", [n for n, _ in _nets()])
def test_mid_equals_oracle_bitwise(Engine, oracle_mod, name):
    from bayesiannetwork_amd import Evidence, synth
    g = dict(_nets())[name]
    with Engine(g) as eng:
        eng.set_option("dag", 0)   # (the k = 4 networks of this list would by default take the register-resident DAG path, bn_dag.hip)
        assert eng.info("small_eligible") == 0 and eng.info("mid_eligible") == 1 and eng.info("mid_parts") >= 2
        for ev, eps, cap in ((Evidence.none(), 1e-6, 0), (synth.random_evidence(g, 0.1, seed=3), 1e-9, 0), (synth.random_evidence(g, 0.3, seed=5), 1e-3, 0),
                             (synth.random_evidence(g, 0.05, seed=6), 1e-12, 3)):
            o = oracle_mod.bp_run(g, ev, eps, cap, dump_msgs=True)
            for _ in range(3):   # repeated runs: nothing of one run leaks into the next
                r = eng.bp_run(ev, eps, cap)
                assert eng.last_path() == 4 and eng.bp_stats()["sweep_launches"] == 1 and eng.info("mid_aborts") == 0
                assert r["sweeps"] == o["sweeps"] and np.array_equal(r["beliefs"], o["beliefs"], equal_nan=True)
                assert np.array_equal(eng.bp_residuals(), o["residuals"]) and r["residual"] == o["residuals"][-1]
                pi, lam = eng.bp_messages()
                assert np.array_equal(pi, o["pi_msg"], equal_nan=True) and np.array_equal(lam, o["lambda_msg"], equal_nan=True)
        # the tile kernels on the same engine, alternating with this path: agree to rounding; staged evidence survives the switch
        ev = synth.random_evidence(g, 0.1, seed=3)
        o = oracle_mod.bp_run(g, ev, 1e-6)
        eng.bp_set_evidence(ev)
        for mid, path in ((1, 4), (0, 0), (1, 4)):
            eng.set_option("mid", mid)
            r = eng.bp_run_device(1e-6)
            assert eng.last_path() == path and r["sweeps"] == o["sweeps"]
            bel = eng.bp_beliefs()
            assert np.array_equal(bel, o["beliefs"]) if mid else np.allclose(bel, o["beliefs"], rtol=0, atol=1e-12)
        eng.set_option("multisweep", 0)      # "one launch per sweep" switches this path off as well
        eng.bp_run(ev, 1e-6)
        assert eng.last_path() == 0


def test_mid_soft_and_zero_evidence_and_a_long_run(Engine, oracle_mod):
    from bayesiannetwork_amd import Evidence, synth
    g = synth.random_dag(80, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=10)
    k = g.k
    soft = Evidence.from_dict(g, {3: np.full(int(k[3]), 1.0 / k[3]), 40: np.arange(1, int(k[40]) + 1, dtype=float), 70: 0})
    zero = Evidence.from_dict(g, {5: np.zeros(int(k[5]))})   # 0/0 -> NaN in the reference (no zero guard, :298-311)
    with Engine(g) as eng:
        for ev, eps, cap in ((soft, 1e-9, 0), (zero, 1e-6, 6)):
            o = oracle_mod.bp_run(g, ev, eps, cap)
            r = eng.bp_run(ev, eps, cap)
            assert eng.last_path() == 4 and r["sweeps"] == o["sweeps"] and np.array_equal(r["beliefs"], o["beliefs"], equal_nan=True)
        assert np.isnan(eng.bp_run(zero, 1e-6, 6)["beliefs"]).any()
        # a launch executes at most 65 536 iterations; the run goes on in another launch from the state in memory
        ev = synth.random_evidence(g, 0.05, seed=1)
        o = oracle_mod.bp_run(g, ev, 0.0, 66000, res_cap=66000)
        r = eng.bp_run(ev, 0.0, 66000)
        assert eng.last_path() == 4 and eng.bp_stats()["sweep_launches"] == 2
        assert r["sweeps"] == 66000 and np.array_equal(r["beliefs"], o["beliefs"])


def test_mid_batches_and_the_view(Engine, oracle_mod):
    """bn_bp_run_batch: every set runs like a single query (same kernel, same bits), as many sets per launch as fit the chip;
    bn_bp_run_view (the drop-in's host path) on this path."""
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import BeliefPropagation
    g = synth.random_dag(300, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=12)
    evs = [synth.random_evidence(g, f, seed=20 + q) for q, f in enumerate([0.0, 0.05, 0.1, 0.3, 0.02, 0.5, 0.2] * 10)]   # 70 sets: three launches
    with Engine(g) as eng:
        for sets, eps, cap in ((evs, 1e-6, 0), (evs[:3], 1e-12, 4), (evs[:1], 1e-6, 0)):
            out = eng.bp_run_batch(sets, eps, cap)
            assert eng.last_path() == 4 and eng.info("mid_aborts") == 0
            for q, ev in enumerate(sets):
                o = oracle_mod.bp_run(g, ev, eps, cap)
                assert out["sweeps"][q] == o["sweeps"] and np.array_equal(out["beliefs"][q], o["beliefs"], equal_nan=True), q
                assert np.array_equal(eng.bp_residuals_batch(q), o["residuals"]), q
        assert len(set(eng.bp_run_batch(evs, 1e-6)["sweeps"].tolist())) > 1
        eng.set_option("mid", 0)             # staged for this path, run on the tile kernels
        eng.bp_set_evidence_batch(evs[:5])
        out = eng.bp_run_batch_device(1e-6)
        assert eng.last_path() == 0
        for q in range(5):
            assert np.allclose(eng.bp_beliefs_batch()[q], oracle_mod.bp_run(g, evs[q], 1e-6)["beliefs"], rtol=0, atol=1e-12)
    bp = BeliefPropagation(g)
    for q in range(40):
        got = np.concatenate([np.asarray(m).ravel() for m in bp(evs[q % 7], 1e-6)])
        assert np.array_equal(got, oracle_mod.bp_run(g, evs[q % 7], 1e-6)["beliefs"]), q
    assert bp.engine.last_path() == 4


def test_mid_batch_on_a_network_of_more_than_100_workgroups(Engine, oracle_mod):
    """108 workgroups per set: two sets per launch fit the chip; five sets = three launches, every set the oracle bit for bit."""
    from bayesiannetwork_amd import synth
    g = synth.random_dag(2000, 4, 64, [2, 3, 4, 3, 2, 4, 4], seed=9)
    evs = [synth.random_evidence(g, f, seed=40 + q) for q, f in enumerate([0.0, 0.05, 0.2, 0.01, 0.1])]
    with Engine(g) as eng:
        assert eng.info("mid_eligible") == 1 and eng.info("mid_parts") > 100
        eng.set_option("dag", 0)   # (arities 2..4 with <= 4 parents: the register-resident DAG path would take it by default)
        out = eng.bp_run_batch(evs, 1e-6)
        assert eng.last_path() == 4 and eng.info("mid_aborts") == 0
        for q, ev in enumerate(evs):
            o = oracle_mod.bp_run(g, ev, 1e-6)
            assert out["sweeps"][q] == o["sweeps"] and np.array_equal(out["beliefs"][q], o["beliefs"], equal_nan=True), q
            assert np.array_equal(eng.bp_residuals_batch(q), o["residuals"]), q

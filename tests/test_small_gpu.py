"""Small networks (ALARM-sized: what the reference's users load): the whole run in ONE workgroup with the state in
LDS (csrc/bn_small.hip, bn_bp_last_path == 3).  Sums and products keep the reference's order for any table size,
so the bar is the oracle BIT FOR BIT -- marginals, sweep count, per-sweep maximum_difference, final messages -- also
on networks with three and more parents per node, where the tile kernels only agree to rounding."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def Engine(bnlib):
    from bayesiannetwork_amd.engine import Engine
    return Engine


def _alarm():
    from bayesiannetwork_amd.dsc import load_dsc
    return load_dsc(os.path.join(HERE, "golden", "alarm_shaped.dsc"))[0]


def _nets():
    from bayesiannetwork_amd import synth
    return [("alarm_shaped", _alarm()), ("pearl", synth.pearl()), ("resume_chain", synth.resume_chain()),
            ("mixed12", synth.random_dag(12, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=2)),
            ("mixed27", synth.random_dag(27, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=4)),
            ("mixed37_4parents", synth.random_dag(37, 4, 16, [2, 3, 4, 3, 2, 4, 5], seed=5)),   # a 360-entry table: 180-term runs
            ("mixed60", synth.random_dag(60, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=9)),            # 16 waves
            ("k7", synth.random_dag(16, 2, 8, [7, 5, 6, 2], seed=8)),                           # arities above 4
            ("grid8", synth.grid(8, 8, 4, seed=1)),                                             # four rounds of items per thread
            ("chain40", synth.grid(40, 1, 3, seed=3))]


def _same_as_oracle(r, hist, msgs, o):
    assert r["sweeps"] == o["sweeps"]
    assert np.array_equal(r["beliefs"], o["beliefs"], equal_nan=True)
    assert np.array_equal(hist, o["residuals"])
    assert r["residual"] == o["residuals"][-1]
    if msgs is not None:
        assert np.array_equal(msgs[0], o["pi_msg"], equal_nan=True) and np.array_equal(msgs[1], o["lambda_msg"], equal_nan=True)


@pytest.mark.parametrize("name", [n for n, _ in _nets()])
def test_small_equals_oracle_bitwise(Engine, oracle_mod, name):
    from bayesiannetwork_amd import Evidence, synth
    g = dict(_nets())[name]
    with Engine(g) as eng:
        eng.set_option("small", 2)   # this file is about the one-workgroup path: wherever eligible (by default a network of several rounds of entry items takes the register-resident DAG path, bn_engine.cpp dag_applies)
        assert eng.info("small_eligible") == 1 and 1 <= eng.info("small_waves") <= 16
        for ev, eps in ((Evidence.none(), 1e-6), (synth.random_evidence(g, 0.1, seed=3), 1e-9), (synth.random_evidence(g, 0.3, seed=5), 1e-3)):
            o = oracle_mod.bp_run(g, ev, eps, dump_msgs=True)
            for _ in range(3):   # repeated runs: nothing of one run leaks into the next
                r = eng.bp_run(ev, eps)
                assert eng.last_path() == 3 and eng.bp_stats()["sweep_launches"] == 1
                _same_as_oracle(r, eng.bp_residuals(), eng.bp_messages(), o)
        # the other paths on the same engine, alternating with this one: agree to rounding (bit for bit with <= 2 parents)
        ev = synth.random_evidence(g, 0.1, seed=3)
        o = oracle_mod.bp_run(g, ev, 1e-6)
        eng.set_option("small", 0)
        r0 = eng.bp_run(ev, 1e-6)
        assert eng.last_path() != 3 and r0["sweeps"] == o["sweeps"]
        assert np.allclose(r0["beliefs"], o["beliefs"], rtol=0, atol=1e-12)
        eng.set_option("small", 2)
        r = eng.bp_run(ev, 1e-6)
        assert eng.last_path() == 3 and np.array_equal(r["beliefs"], o["beliefs"])
        eng.set_option("multisweep", 0)      # "one launch per sweep" switches this path off as well
        eng.bp_run(ev, 1e-6)
        assert eng.last_path() == 0


def test_small_soft_zero_evidence_and_caps(Engine, oracle_mod):
    from bayesiannetwork_amd import Evidence
    g = _alarm()
    k = g.k
    soft = Evidence.from_dict(g, {3: np.full(int(k[3]), 1.0 / k[3]), 10: np.arange(1, int(k[10]) + 1, dtype=float), 20: 0})
    zero = Evidence.from_dict(g, {5: np.zeros(int(k[5]))})   # 0/0 -> NaN in the reference (no zero guard, :298-311)
    with Engine(g) as eng:
        eng.set_option("small", 2)   # this file is about the one-workgroup path: wherever eligible (by default a network of several rounds of entry items takes the register-resident DAG path, bn_engine.cpp dag_applies)
        for ev, eps, cap in ((soft, 1e-9, 0), (zero, 1e-6, 6), (soft, 1e-12, 1), (soft, 1e-12, 2), (soft, 1e-12, 5)):
            o = oracle_mod.bp_run(g, ev, eps, cap, dump_msgs=True)
            r = eng.bp_run(ev, eps, cap)
            assert eng.last_path() == 3
            if cap:
                assert r["sweeps"] <= cap
            assert r["sweeps"] == o["sweeps"]
            assert np.array_equal(r["beliefs"], o["beliefs"], equal_nan=True)
            assert np.array_equal(eng.bp_residuals(), o["residuals"])
            pi, lam = eng.bp_messages()
            assert np.array_equal(pi, o["pi_msg"], equal_nan=True) and np.array_equal(lam, o["lambda_msg"], equal_nan=True)
        assert np.isnan(eng.bp_run(zero, 1e-6, 6)["beliefs"]).any()


def test_small_run_longer_than_one_launch(Engine, oracle_mod):
    """A launch executes at most 65 536 iterations; a longer run continues in another launch from the state the first one
    left in memory."""
    from bayesiannetwork_amd import synth
    g = synth.pearl()
    ev = synth.random_evidence(g, 0.0, seed=1)
    with Engine(g) as eng:
        eng.set_option("small", 2)   # this file is about the one-workgroup path: wherever eligible (by default a network of several rounds of entry items takes the register-resident DAG path, bn_engine.cpp dag_applies)
        o = oracle_mod.bp_run(g, ev, 0.0, 70000, res_cap=70000)   # eps = 0: never converges, stopped at max_sweeps
        r = eng.bp_run(ev, 0.0, 70000)
        assert eng.last_path() == 3 and eng.bp_stats()["sweep_launches"] == 2
        assert r["sweeps"] == 70000 == o["sweeps"] and np.array_equal(r["beliefs"], o["beliefs"])
        hist = eng.bp_residuals()
        assert np.array_equal(hist, o["residuals"][:len(hist)])


def test_small_batch_one_workgroup_per_set(Engine, oracle_mod):
    """bn_bp_run_batch on a small network: one workgroup per evidence set, all sets in one launch, every set stopping on
    its own sweep -- each set bit-identical to the oracle (= to running it alone)."""
    from bayesiannetwork_amd import Evidence, synth
    g = _alarm()
    evs = [synth.random_evidence(g, f, seed=20 + q) for q, f in enumerate([0.0, 0.05, 0.1, 0.3, 0.02, 0.5, 0.2])]
    evs.append(Evidence.from_dict(g, {4: np.full(int(g.k[4]), 0.5)}))
    with Engine(g) as eng:
        eng.set_option("small", 2)   # this file is about the one-workgroup path: wherever eligible (by default a network of several rounds of entry items takes the register-resident DAG path, bn_engine.cpp dag_applies)
        for sets, eps, cap in ((evs, 1e-6, 0), (evs[:3], 1e-12, 4), (evs[:1], 1e-6, 0), (evs * 8, 1e-9, 0), (evs * 32, 1e-6, 0)):   # up to 256 sets: one workgroup per CU
            out = eng.bp_run_batch(sets, eps, cap)
            assert eng.last_path() == 3 and eng.bp_stats()["sweep_launches"] == 1
            for q, ev in enumerate(sets):
                o = oracle_mod.bp_run(g, ev, eps, cap)
                assert out["sweeps"][q] == o["sweeps"], q
                assert np.array_equal(out["beliefs"][q], o["beliefs"], equal_nan=True), q
                assert np.array_equal(eng.bp_residuals_batch(q), o["residuals"]), q
            if cap == 0 and len(sets) > 3:
                assert len(set(out["sweeps"].tolist())) > 1
        # single queries and batches alternate on one engine
        r = eng.bp_run(evs[1], 1e-6)
        assert np.array_equal(r["beliefs"], oracle_mod.bp_run(g, evs[1], 1e-6)["beliefs"])


def test_small_through_the_view_and_evidence_changes(Engine, oracle_mod):
    """The host path of the drop-in (bn_bp_run_view: evidence from mapped staging memory, marginals written straight into
    the page-locked buffer) on this path, 300 queries with changing evidence (the evidence mark wraps at 255)."""
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import BeliefPropagation
    g = _alarm()
    bp = BeliefPropagation(g)
    want = {}
    for q in range(300):
        ev = synth.random_evidence(g, 0.1, seed=q % 7)
        if q % 7 not in want:
            want[q % 7] = oracle_mod.bp_run(g, ev, 1e-6)["beliefs"]
        got = np.concatenate([np.asarray(m).ravel() for m in bp(ev, 1e-6)])   # a list of 1 x k arrays, one per node
        assert np.array_equal(got, want[q % 7]), q
    assert bp.engine.last_path() == 3


def test_small_evidence_is_read_in_place_and_survives_path_changes(Engine, oracle_mod):
    """On this path the kernel reads the evidence arrays where the caller's call left them (no evidence launch in front of the
    run); the tile buffers get marks and vectors only when another path runs.  Staged once, the evidence stays in force
    whichever path the following runs take."""
    from bayesiannetwork_amd import synth
    g = _alarm()
    ev, ev2 = synth.random_evidence(g, 0.15, seed=31), synth.random_evidence(g, 0.1, seed=32)
    o, o2 = oracle_mod.bp_run(g, ev, 1e-6), oracle_mod.bp_run(g, ev2, 1e-6)
    with Engine(g) as eng:
        eng.set_option("small", 2)   # this file is about the one-workgroup path: wherever eligible (by default a network of several rounds of entry items takes the register-resident DAG path, bn_engine.cpp dag_applies)
        eng.bp_set_evidence(ev)
        for small, path in ((1, 3), (0, 0), (1, 3), (0, 0)):
            eng.set_option("small", small)
            r = eng.bp_run_device(1e-6)
            assert eng.last_path() == path and r["sweeps"] == o["sweeps"]
            assert np.array_equal(eng.bp_beliefs(), o["beliefs"])   # (ALARM's tables take the ordered path of the tile kernels: same bits)
        eng.set_option("small", 0)
        eng.bp_set_evidence(ev2)            # staged while the tile path is selected, run on this one
        eng.set_option("small", 2)
        r = eng.bp_run_device(1e-6)
        assert eng.last_path() == 3 and np.array_equal(eng.bp_beliefs(), o2["beliefs"])
        # batches: staged for one path, run on the other
        sets = [ev, ev2, None, ev]
        want = [o, o2, oracle_mod.bp_run(g, None, 1e-6), o]
        for small_at_set, small_at_run in ((1, 0), (0, 1), (1, 1)):
            eng.set_option("small", small_at_set)
            eng.bp_set_evidence_batch(sets)
            eng.set_option("small", small_at_run)
            out = eng.bp_run_batch_device(1e-6)
            assert eng.last_path() == (3 if small_at_run else 0)
            bel = eng.bp_beliefs_batch()
            for q, w in enumerate(want):
                assert out["sweeps"][q] == w["sweeps"] and np.array_equal(bel[q], w["beliefs"]), (small_at_set, small_at_run, q)
        eng.set_option("small", 2)
        assert np.array_equal(eng.bp_run(ev2, 1e-6)["beliefs"], o2["beliefs"])


def test_small_degenerate_networks(Engine, oracle_mod):
    """No edges at all, a single node, arity-1 nodes, a star with 20 children, 64 states, evidence on every node."""
    from bayesiannetwork_amd import Evidence, synth
    from bayesiannetwork_amd.flat import from_parent_lists
    rng = np.random.default_rng(5)

    def table(kv, rows):
        t = rng.random((rows, kv)) + 0.05
        return (t / t.sum(axis=1, keepdims=True)).ravel().tolist()
    nets = {
        "no_edges": from_parent_lists([2, 3, 4, 5], [[], [], [], []], [table(2, 1), table(3, 1), table(4, 1), table(5, 1)]),
        "single": from_parent_lists([3], [[]], [table(3, 1)]),
        "arity1": from_parent_lists([1, 2, 1, 3], [[], [0], [1], [1, 2]], [[1.0], table(2, 1), table(1, 2), table(3, 2)]),
        "star20": from_parent_lists([3] + [2] * 20, [[]] + [[0]] * 20, [table(3, 1)] + [table(2, 3) for _ in range(20)]),
        "k64": from_parent_lists([64, 2, 64], [[], [0], [1]], [table(64, 1), table(2, 64), table(64, 2)]),
    }
    for name, g in nets.items():
        full = Evidence.from_dict(g, {v: int(v % g.k[v]) for v in range(g.n)})
        some = synth.random_evidence(g, 0.3, seed=2)
        with Engine(g) as eng:
            eng.set_option("small", 2)   # this file is about the one-workgroup path: wherever eligible (by default a network of several rounds of entry items takes the register-resident DAG path, bn_engine.cpp dag_applies)
            assert eng.info("small_eligible") == 1, name
            for ev in (Evidence.none(), some, full):
                o = oracle_mod.bp_run(g, ev, 1e-9, dump_msgs=True)
                r = eng.bp_run(ev, 1e-9)
                assert eng.last_path() == 3, name
                assert r["sweeps"] == o["sweeps"] and np.array_equal(r["beliefs"], o["beliefs"], equal_nan=True), name
                assert np.array_equal(eng.bp_residuals(), o["residuals"]), name
                pi, lam = eng.bp_messages()
                assert np.array_equal(pi, o["pi_msg"], equal_nan=True) and np.array_equal(lam, o["lambda_msg"], equal_nan=True), name

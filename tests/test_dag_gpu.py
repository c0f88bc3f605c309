"""k = 4 networks with up to 5 parents per node (BASELINE configs[1], the 10 k-node random DAG): child tiles with the CPT in
registers + parent items on waves of their own, state in device memory, one launch per run (csrc/bn_dag.hip,
bn_bp_last_path == 5).  Nodes with <= 2 parents keep the reference's operation order (bit-identical to the oracle on networks
made of such nodes); with >= 3 parents the contraction is factored: <= 1e-12 against the oracle (whose products over >= 3
parents follow one fixed order where the reference's own follow an unordered_map's, belief_propagation.hpp:253), equal sweep
counts everywhere."""
import numpy as np
import pytest

from helpers import hub_network, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine(bnlib):
    from bayesiannetwork_amd.engine import Engine
    return Engine


def _nets():
    from bayesiannetwork_amd import synth
    return [("grid12", synth.grid(12, 12, 4, seed=3), True),             # <= 2 parents: the reference's order, bit for bit
            ("chain300", synth.random_dag(300, 1, 1, 4, seed=4), True),
            ("grid40", synth.grid(40, 40, 4, seed=5), True),
            ("dag300", synth.random_dag(300, 4, 32, 4, seed=5), False),  # lane groups of 4 and 16
            ("dag200_p5", synth.random_dag(200, 5, 32, 4, seed=6), False),  # ... and 64 (5 parents: 4 096-entry tables)
            ("dag3000", synth.random_dag(3000, 4, 64, 4, seed=8), False),   # nodes with up to ~9 children
            ("hub20", hub_network(20), True),       # a node's 21 parent items share a wave and hand each other their records
            ("hub70", hub_network(70), True)]       # more children than a wave has lanes: every item loads all records itself


@pytest.mark.parametrize("name", [n for n, _, _ in _nets()])
def test_dag_path_vs_oracle(Engine, oracle_mod, name):
    from bayesiannetwork_amd import Evidence, synth
    g, exact = {n: (m, x) for n, m, x in _nets()}[name]
    with Engine(g) as eng:
        assert eng.info("dag_eligible") == 1
        eng.set_option("dag", 2)
        for ev, eps, cap in ((Evidence.none(), 1e-6, 0), (synth.random_evidence(g, 0.1, seed=3), 1e-9, 0),
                             (synth.random_evidence(g, 0.3, seed=5), 1e-3, 0), (synth.random_evidence(g, 0.05, seed=6), 1e-12, 3)):
            o = oracle_mod.bp_run(g, ev, eps, cap, dump_msgs=True)
            first = None
            for _ in range(3):   # repeated runs: nothing of one run leaks into the next, and the bits repeat
                r = eng.bp_run(ev, eps, cap)
                assert eng.last_path() == 5 and eng.bp_stats()["sweep_launches"] == 1 and eng.info("dag_aborts") == 0
                assert r["sweeps"] == o["sweeps"]
                pi, lam = eng.bp_messages()
                if exact:
                    assert np.array_equal(r["beliefs"], o["beliefs"])
                    assert np.array_equal(eng.bp_residuals(), o["residuals"]) and r["residual"] == o["residuals"][-1]
                    assert np.array_equal(pi, o["pi_msg"]) and np.array_equal(lam, o["lambda_msg"])
                else:
                    assert np.abs(r["beliefs"] - o["beliefs"]).max() < 1e-12
                    assert np.abs(eng.bp_residuals() - o["residuals"]).max() < 1e-12
                    assert np.abs(pi - o["pi_msg"]).max() < 1e-12 and np.abs(lam - o["lambda_msg"]).max() < 1e-12
                if first is None:
                    first = r["beliefs"].copy()
                assert np.array_equal(first, r["beliefs"])
        # the other paths on the same engine, alternating with this one: staged evidence survives the switch
        ev = synth.random_evidence(g, 0.1, seed=3)
        o = oracle_mod.bp_run(g, ev, 1e-6)
        eng.bp_set_evidence(ev)
        for dag, want5 in ((2, True), (0, False), (2, True)):
            eng.set_option("dag", dag)
            r = eng.bp_run_device(1e-6)
            assert (eng.last_path() == 5) == want5 and r["sweeps"] == o["sweeps"]
            assert np.abs(eng.bp_beliefs() - o["beliefs"]).max() < 1e-12


def test_dag_soft_and_zero_evidence_and_a_two_launch_run(Engine, oracle_mod):
    from bayesiannetwork_amd import Evidence, synth
    g = synth.random_dag(300, 4, 32, 4, seed=5)
    soft = Evidence.from_dict(g, {3: np.full(4, 0.25), 40: np.arange(1.0, 5.0), 70: 0, 299: 2})
    zero = Evidence.from_dict(g, {5: np.zeros(4)})   # 0/0 -> NaN in the reference (no zero guard, :298-311)
    with Engine(g) as eng:
        eng.set_option("dag", 2)
        for ev, eps, cap in ((soft, 1e-9, 0), (zero, 1e-6, 6)):
            o = oracle_mod.bp_run(g, ev, eps, cap)
            r = eng.bp_run(ev, eps, cap)
            assert eng.last_path() == 5 and r["sweeps"] == o["sweeps"]
            assert rel_err(r["beliefs"], o["beliefs"]) < 1e-9      # (NaNs must coincide)
        assert np.isnan(eng.bp_run(zero, 1e-6, 6)["beliefs"]).any()
        # a launch executes at most 1 024 iterations; the run goes on in another launch from the state in memory
        ev = synth.random_evidence(g, 0.05, seed=1)
        o = oracle_mod.bp_run(g, ev, 0.0, 1100, res_cap=1100)
        r = eng.bp_run(ev, 0.0, 1100)
        assert eng.last_path() == 5 and eng.bp_stats()["sweep_launches"] == 2
        assert r["sweeps"] == 1100 and np.abs(r["beliefs"] - o["beliefs"]).max() < 1e-12
        assert np.abs(eng.bp_residuals()[:1100] - o["residuals"]).max() < 1e-12


@pytest.mark.parametrize("name", ["grid40", "dag300", "dag3000", "hub70"])
def test_dag_dataflow_form(Engine, oracle_mod, name):
    """Option "dagflow" 1: the single query WITHOUT a grid barrier -- a tile waits for its neighbour tiles' granules, a service block
    takes the stop decision one iteration behind (csrc/bn_dag.hip, dag_flow_drive).  One speculative iteration per run, written to
    the other buffer: the same sweep count, residual history, messages and marginals as the barrier form bit for bit (and as the
    oracle on networks of <= 2-parent nodes), also with a sweep cap, a run beyond one launch's budget, NaNs and repeated runs."""
    from bayesiannetwork_amd import Evidence, synth
    g, exact = {n: (m, x) for n, m, x in _nets()}[name]
    with Engine(g) as eng:
        eng.set_option("dag", 2)
        eng.bp_run(None, 1e-3)   # (the path is set up by its first use: eligibility of the dataflow form is known after it)
        if not eng.info("dag_flow_eligible"):
            assert eng.info("dag_blocks") < 2 or eng.info("dag_flow_max_nbr") > 64
            pytest.skip("one block, or a tile with more than 64 neighbour tiles: the barrier form is the only one")
        cases = [(Evidence.none(), 1e-6, 0), (synth.random_evidence(g, 0.1, seed=3), 1e-9, 0), (synth.random_evidence(g, 0.05, seed=6), 1e-12, 3),
                 (synth.random_evidence(g, 0.05, seed=6), 1e-12, 1), (Evidence.from_dict(g, {5: np.zeros(4)}), 1e-6, 6)]
        for ev, eps, cap in cases:
            o = oracle_mod.bp_run(g, ev, eps, cap, dump_msgs=True)
            got = {}
            for flow in (0, 1, 1):
                eng.set_option("dagflow", flow)
                r = eng.bp_run(ev, eps, cap)
                assert eng.last_path() == 5 and eng.info("last_dag_flow") == flow and eng.info("dag_aborts") == 0
                assert eng.bp_stats()["sweep_launches"] == 1
                pi, lam = eng.bp_messages()
                got[flow] = (r["sweeps"], r["residual"], r["beliefs"].copy(), eng.bp_residuals().copy(), pi, lam)
            b, f = got[0], got[1]
            assert b[0] == f[0] == o["sweeps"]
            assert b[1] == f[1] or (np.isnan(b[1]) and np.isnan(f[1]))
            for x, y in zip(b[2:], f[2:]):
                assert np.array_equal(x, y, equal_nan=True)
            if exact:
                assert np.array_equal(f[2], o["beliefs"], equal_nan=True) and np.array_equal(f[3], o["residuals"], equal_nan=True)
                assert np.array_equal(f[4], o["pi_msg"], equal_nan=True) and np.array_equal(f[5], o["lambda_msg"], equal_nan=True)
            else:
                assert rel_err(f[2], o["beliefs"]) < 1e-9
        # beyond one launch's budget (1 024 iterations): the second launch continues from the state in memory, no speculation across the cut
        ev = synth.random_evidence(g, 0.05, seed=1)
        eng.set_option("dagflow", 0)
        want = eng.bp_run(ev, 0.0, 1100)
        want_res = eng.bp_residuals().copy()
        eng.set_option("dagflow", 1)
        r = eng.bp_run(ev, 0.0, 1100)
        assert eng.info("last_dag_flow") == 1 and eng.bp_stats()["sweep_launches"] == 2 and r["sweeps"] == want["sweeps"] == 1100
        assert np.array_equal(r["beliefs"], want["beliefs"]) and np.array_equal(eng.bp_residuals(), want_res)
        # staged evidence, device-resident runs, alternating forms
        eng.bp_set_evidence(synth.random_evidence(g, 0.1, seed=3))
        first = None
        for flow in (1, 0, 1, 1, 0):
            eng.set_option("dagflow", flow)
            rr = eng.bp_run_device(1e-6)
            cur = (rr["sweeps"], eng.bp_beliefs().copy())
            first = first or cur
            assert cur[0] == first[0] and np.array_equal(cur[1], first[1])


def test_dag_dataflow_form_at_config2_size(Engine, oracle_mod):
    """... on BASELINE configs[1] itself (1 189 tiles on 224 blocks + the service block, up to 40 neighbour tiles per tile): the same bits as
    the barrier form; the default stays the barrier form (the dataflow form is the slower one there: EXPERIMENTS R6.2)."""
    from bayesiannetwork_amd import synth
    g = synth.random_dag(10000, 4, 64, 4, seed=1)
    ev = synth.random_evidence(g, 0.01, seed=7)
    o = oracle_mod.bp_run(g, ev, 1e-3)
    with Engine(g) as eng:
        r0 = eng.bp_run(ev, 1e-3)
        assert eng.last_path() == 5 and eng.info("last_dag_flow") == 0 and eng.info("dag_flow_eligible") == 1
        eng.set_option("dagflow", 1)
        for _ in range(3):
            r1 = eng.bp_run(ev, 1e-3)
            assert eng.info("last_dag_flow") == 1 and eng.info("dag_aborts") == 0
            assert r1["sweeps"] == r0["sweeps"] == o["sweeps"] and np.array_equal(r1["beliefs"], r0["beliefs"])
        assert np.abs(r1["beliefs"] - o["beliefs"]).max() < 1e-12


def test_dag_view_and_functor(Engine, oracle_mod):
    """bn_bp_run_view (the drop-in's host path: marginals written straight into the mapped host buffer) on this path."""
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import BeliefPropagation
    g = synth.random_dag(1500, 4, 64, 4, seed=31)
    evs = [synth.random_evidence(g, f, seed=20 + q) for q, f in enumerate([0.0, 0.05, 0.1, 0.3, 0.02])]
    bp = BeliefPropagation(g)
    bp.engine.set_option("dag", 2)
    for q in range(15):
        got = np.concatenate([np.asarray(m).ravel() for m in bp(evs[q % 5], 1e-6)])
        o = oracle_mod.bp_run(g, evs[q % 5], 1e-6)
        assert bp.last["sweeps"] == o["sweeps"] and np.abs(got - o["beliefs"]).max() < 1e-12, q
    assert bp.engine.last_path() == 5


def test_dag_batch_is_a_sequence_of_single_queries(Engine, oracle_mod):
    """bn_bp_run_batch on a network this path takes by default: every set is a single query's launch -- same sweep count, same bits,
    its own residual history -- and the single-query evidence staged before the batch is still in force after it."""
    from bayesiannetwork_amd import synth
    g = synth.random_dag(800, 4, 48, 4, seed=41)
    evs = [synth.random_evidence(g, f, seed=30 + q) for q, f in enumerate([0.0, 0.05, 0.2, 0.01, 0.1, 0.3])]
    with Engine(g) as eng:
        singles = [eng.bp_run(ev, 1e-6) for ev in evs]
        assert eng.last_path() == 5
        hists = []
        for ev in evs:
            eng.bp_run(ev, 1e-6)
            hists.append(eng.bp_residuals().copy())
        eng.bp_set_evidence(evs[2])
        out = eng.bp_run_batch(evs, 1e-6)
        assert eng.last_path() == 5 and eng.info("dag_aborts") == 0
        assert len(set(out["sweeps"].tolist())) > 1
        for q, r in enumerate(singles):
            assert out["sweeps"][q] == r["sweeps"] and np.array_equal(out["beliefs"][q], r["beliefs"]), q
            assert np.array_equal(eng.bp_residuals_batch(q), hists[q]), q
            o = oracle_mod.bp_run(g, evs[q], 1e-6)
            assert r["sweeps"] == o["sweeps"] and np.abs(r["beliefs"] - o["beliefs"]).max() < 1e-12
        r = eng.bp_run_device(1e-6)          # the evidence staged by bn_bp_set_evidence before the batch
        assert r["sweeps"] == singles[2]["sweeps"] and np.array_equal(eng.bp_beliefs(), singles[2]["beliefs"])


def test_dag_config2_full_size_default_path(Engine, oracle_mod):
    """BASELINE configs[1] takes this path by default (beyond the item kernels, not covered by the resident tiles)."""
    from bayesiannetwork_amd import synth
    d = synth.random_dag(10000, 4, 64, 4, seed=1)
    ev = synth.random_evidence(d, 0.01, seed=7)
    with Engine(d) as eng:
        assert eng.info("dag_eligible") == 1 and eng.info("dag_stream") == 0 and eng.info("mid_eligible") == 0
        for eps in (1e-3, 1e-6):
            o = oracle_mod.bp_run(d, ev, eps, threads=8)
            r = eng.bp_run(ev, eps)
            assert eng.last_path() == 5 and eng.info("dag_aborts") == 0
            assert r["sweeps"] == o["sweeps"]
            assert rel_err(r["beliefs"], o["beliefs"]) < 1e-9 and np.abs(r["beliefs"] - o["beliefs"]).max() < 1e-12
            assert np.abs(eng.bp_residuals() - o["residuals"]).max() < 1e-12


def test_dag_100k_nodes_stream_form(Engine, oracle_mod):
    """A 100 k-node DAG of the same kind (218 MB of CPTs: beyond the chip's registers): the same kernel, every wave walking
    several tiles per iteration."""
    from bayesiannetwork_amd import synth
    d = synth.random_dag(100000, 4, 64, 4, seed=11)
    ev = synth.random_evidence(d, 0.01, seed=7)
    o = oracle_mod.bp_run(d, ev, 1e-4, threads=8)
    with Engine(d) as eng:
        assert eng.info("dag_eligible") == 1 and eng.info("dag_stream") == 1
        eng.set_option("dag", 2)
        r = eng.bp_run(ev, 1e-4)
        assert eng.last_path() == 5 and eng.info("dag_aborts") == 0
        assert r["sweeps"] == o["sweeps"]
        assert rel_err(r["beliefs"], o["beliefs"]) < 1e-9 and np.abs(r["beliefs"] - o["beliefs"]).max() < 1e-12
        eng.set_option("dag", 0)
        r0 = eng.bp_run(ev, 1e-4)
        assert eng.last_path() == 0 and r0["sweeps"] == o["sweeps"] and np.abs(r0["beliefs"] - r["beliefs"]).max() < 1e-12


def test_dag_batch_shares_launches(Engine, oracle_mod):
    """Several sets per launch (the sets take turns inside an iteration; up to 16 share the CPT registers): 19 sets = two launches, sets
    that stop on different sweeps, caps; every set keeps the sweep count, the bits and the residual history of its single run, and a
    set that needs more than one launch's 1 024 iterations is finished on its own."""
    from bayesiannetwork_amd import synth
    g = synth.random_dag(600, 4, 48, 4, seed=43)
    evs = [synth.random_evidence(g, f, seed=50 + q) for q, f in enumerate([0.0, 0.05, 0.2, 0.01, 0.1, 0.3, 0.02, 0.15, 0.0, 0.4, 0.07,
                                                                          0.03, 0.25, 0.0, 0.12, 0.06, 0.35, 0.01, 0.09])]
    with Engine(g) as eng:
        singles, hists = [], []
        for ev in evs:
            singles.append(eng.bp_run(ev, 1e-7))
            hists.append(eng.bp_residuals().copy())
        assert eng.last_path() == 5
        out = eng.bp_run_batch(evs, 1e-7)
        assert eng.last_path() == 5 and eng.info("dag_aborts") == 0 and eng.bp_stats()["sweep_launches"] == 2
        assert len(set(out["sweeps"].tolist())) > 1
        for q, r in enumerate(singles):
            assert out["sweeps"][q] == r["sweeps"] and np.array_equal(out["beliefs"][q], r["beliefs"]), q
            assert np.array_equal(eng.bp_residuals_batch(q), hists[q]) and out["residual"][q] == r["residual"], q
        capped = eng.bp_run_batch(evs[:3], 0.0, 5)
        assert capped["sweeps"].tolist() == [5, 5, 5]
        for q in range(3):
            assert np.array_equal(capped["beliefs"][q], eng.bp_run(evs[q], 0.0, 5)["beliefs"])
        long = eng.bp_run_batch(evs[:2], 0.0, 1030)      # beyond a launch's budget: the sets go on alone
        for q in range(2):
            r = eng.bp_run(evs[q], 0.0, 1030)
            assert long["sweeps"][q] == 1030 == r["sweeps"] and np.array_equal(long["beliefs"][q], r["beliefs"])


def test_dag_batch_rerun_on_staged_evidence(Engine):
    """bn_bp_run_batch_device again and again on ONE bn_bp_set_evidence_batch (what a caller with standing evidence sets does, and what
    bench.py times): a batch that fits one launch leaves its evidence in the state slots between runs (an observed node's vectors are
    carried over by every sweep) -- every run gives the single runs' bits; staging other evidence, fewer sets or more sets than a launch
    holds, and a single query in between, each take effect."""
    from bayesiannetwork_amd import synth
    g = synth.random_dag(700, 4, 48, 4, seed=47)
    evs = [synth.random_evidence(g, f, seed=80 + q) for q, f in enumerate([0.0, 0.05, 0.2, 0.01, 0.1, 0.3, 0.02])]
    other = [synth.random_evidence(g, f, seed=95 + q) for q, f in enumerate([0.1, 0.0, 0.25, 0.04, 0.15, 0.02, 0.3])]
    many = [synth.random_evidence(g, 0.01 * (q % 7), seed=120 + q) for q in range(21)]   # two launches: the slots are shared by the chunks
    with Engine(g) as eng:
        single = lambda ev: eng.bp_run(ev, 1e-6)   # noqa: E731
        want = {id(ev): single(ev) for ev in evs + other + many}
        assert eng.last_path() == 5

        def check(sets, runs):
            eng.bp_set_evidence_batch(sets)
            for _ in range(runs):
                out = eng.bp_run_batch_device(1e-6)
                bel = eng.bp_beliefs_batch()
                assert eng.last_path() == 5 and eng.info("dag_aborts") == 0
                for q, ev in enumerate(sets):
                    assert out["sweeps"][q] == want[id(ev)]["sweeps"] and np.array_equal(bel[q], want[id(ev)]["beliefs"]), q
        check(evs, 3)
        check(other, 2)            # other evidence on the same nodes' slots
        check(evs[:3], 2)          # fewer sets
        check(many, 2)             # more than one launch holds: applied per chunk, every run
        check(evs, 2)              # ... and back to a batch that stays
        assert np.array_equal(single(other[2])["beliefs"], want[id(other[2])]["beliefs"])   # a single query in between (its own state)
        out = eng.bp_run_batch_device(1e-6)
        assert np.array_equal(eng.bp_beliefs_batch()[1], want[id(evs[1])]["beliefs"]) and out["sweeps"][1] == want[id(evs[1])]["sweeps"]


@pytest.mark.parametrize("case", ["stream", "one_block"])
def test_dag_batch_other_forms(Engine, oracle_mod, case):
    """The several-sets launch in the kernel's other two forms: the stream form (waves walking several tiles per iteration and set: a
    15 k-node DAG) and a one-block grid (no granules: the block's last wave decides)."""
    from bayesiannetwork_amd import synth
    g = synth.random_dag(15000, 4, 64, 4, seed=3) if case == "stream" else synth.random_dag(24, 3, 8, 4, seed=12)
    evs = [synth.random_evidence(g, f, seed=70 + q) for q, f in enumerate([0.0, 0.05, 0.2])]
    with Engine(g) as eng:
        eng.set_option("dag", 2)
        assert eng.info("dag_eligible") == 1 and eng.info("dag_stream") == (1 if case == "stream" else 0)
        assert case == "stream" or eng.info("dag_blocks") == 1
        singles = [eng.bp_run(ev, 1e-5) for ev in evs]
        assert eng.last_path() == 5
        out = eng.bp_run_batch(evs, 1e-5)
        assert eng.last_path() == 5 and eng.bp_stats()["sweep_launches"] == 1 and eng.info("dag_aborts") == 0
        for q, r in enumerate(singles):
            assert out["sweeps"][q] == r["sweeps"] and np.array_equal(out["beliefs"][q], r["beliefs"]), q
        o = oracle_mod.bp_run(g, evs[1], 1e-5, threads=8)
        assert out["sweeps"][1] == o["sweeps"] and np.abs(out["beliefs"][1] - o["beliefs"]).max() < 1e-12


def test_dag_batch_after_reload(Engine, oracle_mod):
    """bn_reload_cpt between two batches: the second batch runs on the new tables (register images rebuilt, batch state kept)."""
    from bayesiannetwork_amd import synth
    g = synth.random_dag(500, 4, 32, 4, seed=21)
    g2 = synth.random_dag(500, 4, 32, 4, seed=21)
    rng = np.random.default_rng(5)
    cpt = g2.cpt.reshape(-1, 4).copy()
    cpt[::7] = rng.dirichlet(np.ones(4), size=len(cpt[::7]))
    g2.cpt[:] = cpt.ravel()
    evs = [synth.random_evidence(g, f, seed=80 + q) for q, f in enumerate([0.02, 0.1, 0.0, 0.3])]
    with Engine(g) as eng:
        a = eng.bp_run_batch(evs, 1e-6)
        assert eng.last_path() == 5
        eng.reload_cpt(g2.cpt)
        b = eng.bp_run_batch(evs, 1e-6)
        assert eng.last_path() == 5
    for q, ev in enumerate(evs):
        o1, o2 = oracle_mod.bp_run(g, ev, 1e-6), oracle_mod.bp_run(g2, ev, 1e-6)
        assert a["sweeps"][q] == o1["sweeps"] and np.abs(a["beliefs"][q] - o1["beliefs"]).max() < 1e-12
        assert b["sweeps"][q] == o2["sweeps"] and np.abs(b["beliefs"][q] - o2["beliefs"]).max() < 1e-12


def _mixed_nets():
    from bayesiannetwork_amd import synth
    return [("mix2", synth.random_dag(400, 2, 16, [2, 3, 4], seed=17), True),        # <= 2 parents: the reference's bits
            ("grid3", synth.grid(20, 20, 3, seed=4), True),
            ("binary", synth.random_dag(300, 2, 16, 2, seed=19), True),
            ("mix4", synth.random_dag(600, 4, 32, [2, 3, 4, 2], seed=18), False),     # lane groups over padded tables
            ("mix5", synth.random_dag(150, 5, 32, [3, 2, 4], seed=20), False)]


@pytest.mark.parametrize("name", [n for n, _, _ in _mixed_nets()])
def test_dag_arities_below_four(Engine, oracle_mod, name):
    """Arities 2..4 on the register-resident path: tables padded to
    four states with zeros, the initial state written to memory before a run.  A zero term adds nothing to a sum and a zero factor
    keeps a padding entry at zero, so networks of <= 2-parent nodes equal the oracle bit for bit (marginals, residual history,
    messages); with lane groups <= 1e-12; sweep counts equal.  Single queries, soft evidence, repeated runs, batches."""
    from bayesiannetwork_amd import Evidence, synth
    g, exact = {n: (m, x) for n, m, x in _mixed_nets()}[name]
    soft = Evidence.from_dict(g, {v: np.linspace(0.2, 1.0, g.k[v]) for v in (3, 40, 100)})
    with Engine(g) as eng:
        assert eng.info("dag_eligible") == 1
        # the default: this path where some node has >= 3 parents or at least a quarter of the padded tables is real (a binary
        # network of <= 2-parent nodes uses an eighth of its registers: it stays with the tiles / item kernels)
        assert eng.bp_run(Evidence.none(), 1e-6)["sweeps"] > 0 and (eng.last_path() == 5) == (name != "binary")
        eng.set_option("dag", 2)
        evs = [Evidence.none(), synth.random_evidence(g, 0.1, seed=3), synth.random_evidence(g, 0.3, seed=5), soft]
        singles = []
        for ev in evs:
            o = oracle_mod.bp_run(g, ev, 1e-8, dump_msgs=True)
            for _ in range(2):
                r = eng.bp_run(ev, 1e-8)
                assert eng.last_path() == 5 and eng.info("dag_aborts") == 0 and r["sweeps"] == o["sweeps"]
                pi, lam = eng.bp_messages()
                if exact:
                    assert np.array_equal(r["beliefs"], o["beliefs"]) and np.array_equal(eng.bp_residuals(), o["residuals"])
                    assert np.array_equal(pi, o["pi_msg"]) and np.array_equal(lam, o["lambda_msg"])
                else:
                    assert np.abs(r["beliefs"] - o["beliefs"]).max() < 1e-12 and np.abs(eng.bp_residuals() - o["residuals"]).max() < 1e-12
                    assert np.abs(pi - o["pi_msg"]).max() < 1e-12 and np.abs(lam - o["lambda_msg"]).max() < 1e-12
            singles.append(r)
        out = eng.bp_run_batch(evs, 1e-8)
        assert eng.last_path() == 5
        for q, r in enumerate(singles):
            assert out["sweeps"][q] == r["sweeps"] and np.array_equal(out["beliefs"][q], r["beliefs"]), q
        # a capped run and a run continued past a launch's budget
        o = oracle_mod.bp_run(g, evs[1], 0.0, 1030, res_cap=1030)
        r = eng.bp_run(evs[1], 0.0, 1030)
        assert r["sweeps"] == 1030 and eng.bp_stats()["sweep_launches"] == 2
        assert (np.array_equal(r["beliefs"], o["beliefs"]) if exact else np.abs(r["beliefs"] - o["beliefs"]).max() < 1e-12)


def test_dag_hubs_near_and_beyond_the_child_bound(Engine, oracle_mod):
    """A node's parent items carry `child count | target's rank << 16` in one signed word (bn_dag.hpp), and a node with more than
    63 children costs deg^2 record loads per sweep: the plan takes hubs up to kDagMaxChildren = 1 024 children -- every rank intact,
    the oracle's bits (<= 2 parents per node) -- and refuses larger ones, which run on the other paths with the same result."""
    from bayesiannetwork_amd import Evidence, synth
    for nch in (200, 1000):
        g = hub_network(nch)
        with Engine(g) as eng:
            assert eng.info("dag_eligible") == 1
            eng.set_option("dag", 2)
            for ev, eps in ((Evidence.none(), 1e-6), (synth.random_evidence(g, 0.05, seed=3), 1e-9)):
                o = oracle_mod.bp_run(g, ev, eps, dump_msgs=True)
                r = eng.bp_run(ev, eps)
                assert eng.last_path() == 5 and r["sweeps"] == o["sweeps"]
                # (1 000 lambda-messages multiplied into lambda(hub) underflow to 0 and normalise to 0/0: the reference's NaNs, in the same places)
                assert np.array_equal(r["beliefs"], o["beliefs"], equal_nan=True) and np.array_equal(eng.bp_residuals(), o["residuals"])
                pi, lam = eng.bp_messages()
                assert np.array_equal(pi, o["pi_msg"], equal_nan=True) and np.array_equal(lam, o["lambda_msg"], equal_nan=True)
    g = hub_network(1025)
    with Engine(g) as eng:
        assert eng.info("dag_eligible") == 0
        eng.set_option("dag", 2)   # asking for it changes nothing: the network is not eligible
        ev = synth.random_evidence(g, 0.05, seed=3)
        o = oracle_mod.bp_run(g, ev, 1e-9)
        r = eng.bp_run(ev, 1e-9)
        assert eng.last_path() != 5 and r["sweeps"] == o["sweeps"] and rel_err(r["beliefs"], o["beliefs"]) < 1e-9   # (NaNs must coincide)

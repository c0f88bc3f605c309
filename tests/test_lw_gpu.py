"""GPU parity of likelihood weighting (likelihood_weighting.hpp) through the C ABI.

Integer work (sampled states) must be bit-identical to the oracle, which shares the kernel's
Philox-seeded xoshiro128++ streams; weights and histograms agree to fp64 summation order (1e-9 relative).
Against the reference itself parity is statistical: its engine is seeded from
std::random_device (:224-244), so the golden holds a reseeded 1e5-sample run and exact marginals."""
import numpy as np
import pytest

from helpers import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine(bnlib):
    from bayesiannetwork_amd.engine import Engine
    return Engine


@pytest.mark.parametrize("n,maxp,k,frac,ns", [
    (300, 3, [2, 3, 4], 0.05, 5000), (1000, 4, 4, 0.02, 3000), (64, 2, 9, 0.1, 2048), (500, 4, 2, 0.0, 4096),
    # nodes with 5 parents (1 024-row tables: thresholds gathered from memory) among nodes whose tables go through LDS; binary nodes with
    # up to 6 parents (parent lists beyond the four inline ones)
    (400, 5, [2, 3, 4], 0.03, 3000), (200, 6, 2, 0.05, 3000),
])
def test_lw_states_bit_exact_vs_oracle(Engine, oracle_mod, n, maxp, k, frac, ns):
    """Which sampling kernel runs follows from the network: <= 4 parents, <= 256 rows and <= 4 states everywhere -> the straight-line
    kernel (arities all powers of two or not: two instantiations); anything else -> the generic one."""
    from bayesiannetwork_amd import synth
    d = synth.random_dag(n, maxp, 32, k, seed=n + 1)
    ev = synth.random_evidence(d, frac, seed=4).hard_states(d)
    want = oracle_mod.lw_run(d, ev, ns, seed=1234, s_begin=77, states_cap=ns)
    with Engine(d) as eng:
        hist = eng.lw_run(ev, ns, seed=1234, sample_begin=77)
        states, weights = eng.lw_states(ns)
    assert np.array_equal(states, want["states"]), "sampled states are integer work: bit-exact"
    assert np.allclose(weights, want["weights"], rtol=1e-12, atol=0)
    assert np.allclose(hist, want["hist"], rtol=1e-9, atol=1e-12)
    if frac == 0.0:  # unit weights: histograms are integer counts, exact in fp64
        assert np.array_equal(hist, want["hist"])


@pytest.mark.parametrize("n,maxp,k", [(300, 3, [2, 3, 4]), (600, 4, 4)])
def test_lw_generic_kernel_on_a_small_arity_network(Engine, oracle_mod, monkeypatch, n, maxp, k):
    """BN_LW_SMALL=0 sends a network the straight-line kernel would take through the generic kernel (its LDS-staged thresholds, its
    byte-packed row numbers): the same states bit for bit, from both."""
    from bayesiannetwork_amd import synth
    d = synth.random_dag(n, maxp, 32, k, seed=n + 3)
    ev = synth.random_evidence(d, 0.04, seed=5).hard_states(d)
    want = oracle_mod.lw_run(d, ev, 4096, seed=99, s_begin=5, states_cap=4096)
    got = []
    for small in ("1", "0"):
        monkeypatch.setenv("BN_LW_SMALL", small)
        with Engine(d) as eng:
            hist = eng.lw_run(ev, 4096, seed=99, sample_begin=5)
            states, weights = eng.lw_states(4096)
        assert np.array_equal(states, want["states"]) and np.allclose(weights, want["weights"], rtol=1e-12, atol=0)
        assert np.allclose(hist, want["hist"], rtol=1e-9, atol=1e-12)
        got.append(states)
    assert np.array_equal(got[0], got[1])


def test_lw_config5_full_size(Engine, oracle_mod):
    """BASELINE.json configs[4] at its workload: the 10 k-node DAG with 1 % evidence.
    (a) integer work: the first 2048 sampled states are bit-equal to the oracle;
    (b) a 2 M-sample batch on the GPU (the bench step is 10 M: test_lw_config5_ten_million_samples below): split-range additivity at that size, and a
        subset of the same sample ids re-drawn in isolation equals the oracle's histogram of those ids
        (the oracle draws 3e3 samples/s, so the subset is what it finishes in seconds);
    (c) size-independent properties of the 2 M-sample histogram: every node's bins sum to the total
        weight, evidence nodes hold all of it in the observed state.
    Reference path: likelihood_weighting.hpp:28-59."""
    from bayesiannetwork_amd import synth
    d = synth.random_dag(10000, 4, 64, 4, seed=1)
    ev = synth.random_evidence(d, 0.01, seed=7).hard_states(d)
    assert int((ev >= 0).sum()) == 100
    n_big, seed = 2 * 1024 * 1024, 1
    with Engine(d) as eng:
        head = oracle_mod.lw_run(d, ev, 2048, seed=seed, states_cap=2048)
        h_head = eng.lw_run(ev, 2048, seed=seed)
        states, weights = eng.lw_states(2048)
        assert np.array_equal(states, head["states"])
        assert np.allclose(weights, head["weights"], rtol=1e-12, atol=0)
        assert np.allclose(h_head, head["hist"], rtol=1e-9, atol=1e-300)
        whole = eng.lw_run(ev, n_big, seed=seed)
        a = eng.lw_run(ev, n_big // 2, seed=seed)
        b = eng.lw_run(ev, n_big // 2, seed=seed, sample_begin=n_big // 2)
        assert np.allclose(whole, a + b, rtol=1e-9, atol=1e-300)
        # a window of ids from the middle of the big batch, alone, against the oracle on the same ids
        lo, cnt = 1234567, 4096
        sub = eng.lw_run(ev, cnt, seed=seed, sample_begin=lo)
        want = oracle_mod.lw_run(d, ev, cnt, seed=seed, s_begin=lo)
        assert np.allclose(sub, want["hist"], rtol=1e-9, atol=1e-300)
        st_sub, _ = eng.lw_states(64)
        assert np.array_equal(st_sub, oracle_mod.lw_run(d, ev, 64, seed=seed, s_begin=lo, states_cap=64)["states"])
    sums = np.add.reduceat(whole, d.node_off[:-1])
    assert np.allclose(sums, sums[0], rtol=1e-9)
    for v in np.nonzero(ev >= 0)[0]:
        row = whole[d.node_off[v]:d.node_off[v + 1]]
        assert row[ev[v]] > 0 and np.count_nonzero(row) == 1


def test_lw_config5_ten_million_samples(Engine):
    """BASELINE.json configs[4] AT ITS SIZE: one bn_lw_run of 10 M weighted samples on the 10 k-node DAG -- more than the state
    matrix holds at once (32 GiB / 10 000 nodes = 3.4 M samples), so the call walks three batches (bn_lw.cpp lw_run) -- equals the
    sum of four calls over disjoint windows of its sample ids, each of which fits one batch; every node's bins sum to the total
    weight; evidence nodes hold all of it in the observed state.  Reference path: likelihood_weighting.hpp:28-59 (sample_num)."""
    from bayesiannetwork_amd import synth
    d = synth.random_dag(10000, 4, 64, 4, seed=1)
    ev = synth.random_evidence(d, 0.01, seed=7).hard_states(d)
    n_all, seed = 10_000_000, 3
    windows = [(0, 3_000_000), (3_000_000, 3_000_000), (6_000_000, 3_000_000), (9_000_000, 1_000_000)]
    with Engine(d) as eng:
        whole = eng.lw_run(ev, n_all, seed=seed)
        parts = sum(eng.lw_run(ev, cnt, seed=seed, sample_begin=lo) for lo, cnt in windows)
    assert np.allclose(whole, parts, rtol=1e-9, atol=1e-300)
    sums = np.add.reduceat(whole, d.node_off[:-1])
    assert sums[0] > 0 and np.allclose(sums, sums[0], rtol=1e-9)
    for v in np.nonzero(ev >= 0)[0]:
        row = whole[d.node_off[v]:d.node_off[v + 1]]
        assert row[ev[v]] > 0 and np.count_nonzero(row) == 1


def _tie_network(wide):
    """Tables whose running totals sit exactly ON a multiple of 2^-16, one step of the 53-bit grid BELOW and one ABOVE: a draw
    whose top 16 bits equal a threshold's is decided by the 37 bits below (bn_lw_kernels.hip: the tie branch), and those three
    placements are where an off-by-one in that branch would show.  Roots of arity 4 / 3 / 2, children with one and two parents
    (tables staged through LDS), and a binary node with five parents (thresholds gathered from memory by the generic kernel)."""
    from bayesiannetwork_amd import from_parent_lists
    g = 2.0 ** -53

    def row(totals):   # probabilities whose left-to-right running totals are exactly `totals` (all on the 2^-53 grid: exact in fp64)
        t = [0.0] + list(totals) + [1.0]
        return [t[i + 1] - t[i] for i in range(len(t) - 1)]

    j = lambda x: x / 65536.0
    k = [4, 3, 2, 4, 4, 2, 2, 2, 2]
    parents = [[], [], [], [0], [1, 2], [2], [5], [2, 5, 6], [0, 2, 5, 6, 7]]   # node 8: five parents, 4 * 2 * 2 * 2 * 2 = 64 rows, not a packed step
    cpts = [row([j(9000), j(30000) - g, j(50000) + g]), row([j(20000) + g, j(45000)]), row([j(32768) - g]),
            sum((row([j(1000 + 700 * r), j(21000 + 900 * r) + (g if r % 2 else -g), j(60000 - 1100 * r)]) for r in range(4)), []),
            sum((row([j(5000 + 3000 * r) - g, j(33000 + 2000 * r), j(52000 + 1500 * r) + g]) for r in range(6)), []),
            sum((row([j(12345 + 111 * r) + (g if r else 0.0)]) for r in range(2)), []),
            sum((row([j(40000 - 7 * r) - (g if r else 0.0)]) for r in range(2)), []),
            sum((row([j(100 + 8000 * r)]) for r in range(8)), []),
            sum((row([j(300 + 1000 * r) + (g, 0.0, -g)[r % 3]]) for r in range(64)), [])]
    if not wide:   # without the five-parent node every node has <= 4 parents, <= 256 rows, <= 4 states: the straight-line kernel's domain
        k, parents, cpts = k[:8], parents[:8], cpts[:8]
    return from_parent_lists(k=k, parents=parents, cpts=cpts, name="tie_network")


@pytest.mark.parametrize("small,wide", [("1", False), ("0", False), ("1", True)])
def test_lw_draws_that_tie_with_a_threshold(Engine, oracle_mod, monkeypatch, small, wide):
    """make_random_by_weight (likelihood_weighting.hpp:177-193): first i with cum_{i-1} <= u < cum_i.  The kernels decide a draw
    by the top 16 bits of its uniform and fall back to the full 53-bit thresholds when those tie; here the ties are forced
    (thresholds on, just below and just above a multiple of 2^-16) and every state of 400 000 samples must equal the oracle's
    fp64 rule -- from the straight-line kernel, from the generic kernel on the same network (BN_LW_SMALL=0: every table staged
    through LDS), and from the generic kernel with the five-parent node (its thresholds gathered from memory)."""
    monkeypatch.setenv("BN_LW_SMALL", small)
    d = _tie_network(wide)
    ns, seed = 400_000, 20251003
    ev = np.full(d.n, -1, dtype=np.int32)
    want = oracle_mod.lw_run(d, ev, ns, seed=seed, states_cap=ns)
    with Engine(d) as eng:
        hist = eng.lw_run(ev, ns, seed=seed)
        assert eng.info("lw_small") == int(small == "1" and not wide)
        states, _ = eng.lw_states(ns)
    assert np.array_equal(states, want["states"])
    assert np.array_equal(hist, want["hist"])   # unit weights: integer counts
    # the forced ties did occur: samples whose top 16 bits at position 0 (node 0) equal one of its thresholds' top 16 bits
    hits = 0
    for smp in range(0, 120_000):
        h = int(oracle_mod.lw_uniform(seed, smp, 0) * 65536.0)
        hits += h in (9000, 29999, 30000, 50000)
    assert hits >= 1


def test_lw_split_runs_sum(Engine):
    """Histograms of disjoint sample ranges add up (what a multi-GPU reduce relies on)."""
    from bayesiannetwork_amd import synth
    d = synth.random_dag(200, 3, 16, 3, seed=8)
    ev = synth.random_evidence(d, 0.0, seed=1).hard_states(d)
    with Engine(d) as eng:
        whole = eng.lw_run(ev, 6000, seed=5, sample_begin=0)
        a = eng.lw_run(ev, 2500, seed=5, sample_begin=0)
        b = eng.lw_run(ev, 3500, seed=5, sample_begin=2500)
    assert np.array_equal(whole, a + b)


def test_lw_statistical_parity_with_reference(Engine):
    """Pearl net, H=0: z-test of the GPU estimate against exact marginals, and the
    reference's own reseeded 1e5-sample estimate must sit in the same band."""
    from bayesiannetwork_amd.engine import normalize_histogram
    model, _, x = load_golden("lw_pearl")
    n = 400000
    with Engine(model) as eng:
        p = normalize_histogram(model, eng.lw_run(x["ev_state"], n, seed=99))
    exact = x["exact_marginals"]
    # effective sample size under weighting is below n; 5-sigma band with n_eff >= n/4
    band = 5 * np.sqrt(np.maximum(exact * (1 - exact), 1e-12) / (n / 4))
    assert (np.abs(p - exact) <= band + 1e-12).all()
    band_ref = 5 * np.sqrt(np.maximum(exact * (1 - exact), 1e-12) / (1e5 / 4))
    assert (np.abs(x["ref_marginals"] - exact) <= band_ref + 1e-12).all()


def test_lw_dag_vs_reference_estimate(Engine):
    from bayesiannetwork_amd.engine import normalize_histogram
    model, _, x = load_golden("lw_dag30")
    n = 800000
    with Engine(model) as eng:
        p = normalize_histogram(model, eng.lw_run(x["ev_state"], n, seed=3))
    ref = x["ref_marginals"]  # 2e5 reference samples
    band = 6 * np.sqrt(np.maximum(ref * (1 - ref), 1e-4) / (2e5 / 8))
    assert (np.abs(p - ref) <= band).all()


def test_lw_error_paths(Engine):
    from bayesiannetwork_amd import _lib, synth
    d = synth.random_dag(50, 2, 8, 3, seed=2)
    with Engine(d) as eng:
        ev = np.full(d.n, -1, np.int32)
        ev[3] = 7  # state out of range: the reference throws std::out_of_range (:151)
        with pytest.raises(_lib.BnError):
            eng.lw_run(ev, 100, seed=1)


def test_lw_allreduce_one_rank_communicator(Engine):
    """bn_lw_run_allreduce on a 1-rank RCCL communicator equals bn_lw_run (the n-rank split is the
    sample-range arithmetic covered by test_lw_split_runs_sum)."""
    from bayesiannetwork_amd import synth
    d = synth.random_dag(300, 3, 16, 4, seed=4)
    ev = synth.random_evidence(d, 0.03, seed=2).hard_states(d)
    with Engine(d) as eng:
        eng.comm_init(Engine.comm_unique_id())
        a = eng.lw_run(ev, 5000, seed=9, sample_begin=100)
        b = eng.lw_run_allreduce(ev, 5000, seed=9, sample_begin=100)
    assert np.allclose(a, b, rtol=1e-12, atol=0)


@pytest.mark.parametrize("n,maxp,k,ne,want", [(40, 3, [2, 3], 2, 3000), (12, 2, 4, 1, 5000), (200, 4, 2, 0, 1000)])
def test_rejection_sampling_bit_exact_vs_oracle(Engine, oracle_mod, n, maxp, k, ne, want):
    """Counts of accepted samples are integer work: bit-identical to the oracle, as are the
    numbers of samples drawn and accepted (reference rejection_sampling.hpp:93-111 stops at exactly
    `want` accepted samples)."""
    from bayesiannetwork_amd import synth
    d = synth.random_dag(n, maxp, 16, k, seed=n)
    ev = np.full(d.n, -1, np.int32)
    for j in range(ne):
        ev[(7 * j + 3) % d.n] = 0
    wc, wd, wa = oracle_mod.rs_run(d, ev, want, seed=5, s_begin=11, max_draw=1 << 22)
    with Engine(d) as eng:
        c, drawn, acc = eng.rs_run(ev, want, seed=5, max_draw=1 << 22, sample_begin=11)
    assert (drawn, acc) == (wd, wa) and acc == want
    assert np.array_equal(c, wc)
    sums = np.add.reduceat(c, d.node_off[:-1])
    assert (sums == want).all()


def test_rejection_sampling_reference_case_and_cap(Engine):
    """libs/bayesian/test/rejection_sampling.cpp: condition {v4 = 1, v1 = 0}; the exact posterior is
    enumerated here (32 joint states) and the estimate must sit within 5 sigma."""
    from bayesiannetwork_amd import from_parent_lists
    from bayesiannetwork_amd.engine import RejectionSampling
    m = from_parent_lists([2] * 5, [[], [0], [0], [1], [1, 2]],
                          [[.5, .5], [.8, .2, .1, .9], [.7, .3, .4, .6], [.6, .4, .1, .9], [.1, .9, .2, .8, .3, .7, .4, .6]])
    joint = np.zeros([2] * 5)
    for s in np.ndindex(*joint.shape):
        joint[s] = (m.cpt_of(0)[0, s[0]] * m.cpt_of(1)[s[0], s[1]] * m.cpt_of(2)[s[0], s[2]] * m.cpt_of(3)[s[1], s[3]]
                    * m.cpt_of(4)[s[1] * 2 + s[2], s[4]])
    post = joint[0, :, :, 1, :].sum(axis=(1, 2))
    post /= post.sum()
    assert abs(post[0] - 0.62) < 0.062            # the reference test's teacher value, 10 %
    rs = RejectionSampling(m, seed=42)
    n = 200000
    marg = rs({3: 1, 0: 0}, n)
    assert np.abs(marg[1] - post).max() < 5 * np.sqrt(0.25 / n)
    assert marg[3].tolist() == [0.0, 1.0] and marg[0].tolist() == [1.0, 0.0]
    impossible = from_parent_lists([2, 2], [[], [0]], [[1.0, 0.0], [1.0, 0.0, 0.5, 0.5]])
    with pytest.raises(RuntimeError):             # the reference would never return
        RejectionSampling(impossible, max_draws=20000)({1: 1}, 10)


def test_fit_cpt_matches_restatement_and_recovers_cpts(Engine, oracle_mod):
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd._lib import BnError
    # sampler::make_cpt (SURVEY f-3): counting is integer work -> CPTs bit-identical to the C
    # restatement; fitted on many forward samples they approach the generating CPTs; rows no pattern
    # supports are uniform (reference sampler.hpp:140-146).
    d = synth.random_dag(60, 3, 12, [2, 3, 4], seed=5)
    with Engine(d) as eng:
        n = 200000
        eng.lw_run(np.full(d.n, -1, np.int32), n, seed=3)           # forward samples (no evidence)
        states, _ = eng.lw_states(n)
        pats, cnts = np.unique(states, axis=0, return_counts=True)
        fit = eng.fit_cpt(pats, cnts)
        assert np.array_equal(fit, oracle_mod.make_cpt(d, pats, cnts))
        for v in range(d.n):
            rows = fit[d.cpt_off[v]:d.cpt_off[v + 1]].reshape(-1, d.k[v])
            assert np.allclose(rows.sum(axis=1), 1.0, atol=1e-12)
        assert (np.abs(fit - d.cpt) < 0.08).mean() > 0.9              # well-supported rows are recovered
        fit2 = eng.fit_cpt(pats[:1], cnts[:1])                        # one pattern: one row per node is seen
        assert np.array_equal(fit2, oracle_mod.make_cpt(d, pats[:1], cnts[:1]))
        v = int(np.argmax(np.diff(d.in_ptr)))
        rows = fit2[d.cpt_off[v]:d.cpt_off[v + 1]].reshape(-1, d.k[v])
        assert int((rows == 1.0 / d.k[v]).all(axis=1).sum()) == rows.shape[0] - 1
        with pytest.raises(BnError):
            eng.fit_cpt(pats[:1], np.zeros(1, np.uint64))             # sampling_size() == 0 (sampler.hpp:83)


def test_sampler_mirror_make_cpt(Engine, oracle_mod):
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import Sampler
    m = synth.pearl()
    truth = m.cpt.copy()
    with Engine(m) as eng:
        eng.lw_run(np.full(m.n, -1, np.int32), 100000, seed=9)
        states, _ = eng.lw_states(100000)
    pats, cnts = np.unique(states, axis=0, return_counts=True)
    smp = Sampler()
    smp.load_sample({tuple(p): int(c) for p, c in zip(pats.tolist(), cnts.tolist())})
    assert smp.sampling_size() == 100000
    assert smp.make_cpt(m) is True
    assert np.array_equal(m.cpt, oracle_mod.make_cpt(m, pats, cnts))
    assert np.abs(m.cpt - truth).max() < 0.02


def _kahn_min_order(model):
    """The sampler's visiting order (csrc/bn_lw.cpp: Kahn's algorithm, smallest ready node first)."""
    import heapq
    indeg = np.diff(model.in_ptr).astype(int)
    children = [[] for _ in range(model.n)]
    for v in range(model.n):
        for p in model.parents(v):
            children[int(p)].append(v)
    ready = [v for v in range(model.n) if indeg[v] == 0]
    heapq.heapify(ready)
    order = []
    while ready:
        v = heapq.heappop(ready)
        order.append(v)
        for c in children[v]:
            indeg[c] -= 1
            if indeg[c] == 0:
                heapq.heappush(ready, c)
    return np.asarray(order, np.int32)


@pytest.mark.parametrize("case", ["pearl", "dag12", "reversed_chain"])
def test_make_samples_stops_like_the_oracle(Engine, oracle_mod, case):
    """likelihood_weighting::make_samples (likelihood_weighting.hpp:62-117) on the GPU: the units
    executed, the joint-pattern table and the marginals equal the oracle's restatement of that loop fed
    with the GPU's own stream (the same restatement reproduces the reference's reseeded runs bit for bit,
    tests/test_oracle_golden.py) -- so the adaptive stop rule and the pattern table are pinned, not
    just the marginals."""
    from bayesiannetwork_amd import from_parent_lists, synth
    from bayesiannetwork_amd.engine import LikelihoodWeighting
    if case == "pearl":
        model, ev, unit, eps = synth.pearl(), {3: 0}, 20000, 0.004
    elif case == "dag12":
        model = synth.random_dag(12, 3, 6, [2, 3, 2, 2], seed=33)
        ev, unit, eps = {4: 1, 9: 0}, 3000, 0.006   # (the moves per unit are 0.0113, 0.0049999999999994: not an eps a rounding can flip)
    else:  # vertex order is NOT topological: 3 <- 2 <- 1 <- 0 reversed, node 0 is the leaf
        model = from_parent_lists([2, 3, 2, 2], [[1], [2], [3], []],
                                  [[.3, .7, .6, .4, .5, .5], [.2, .3, .5, .6, .3, .1], [.9, .1, .4, .6], [.35, .65]])
        ev, unit, eps = {0: 1}, 5000, 0.006
    ev_state = np.full(model.n, -1, np.int32)
    for v, s in ev.items():
        ev_state[v] = s
    lw = LikelihoodWeighting(model, seed=4242)
    pats, marg = lw.make_samples(ev, unit, eps)
    want = oracle_mod.make_samples(model, ev_state, unit, eps, seed=4242, stream="repo", order=_kahn_min_order(model))
    assert lw.last_units == want["units"] and want["units"] >= 2
    got = sorted(pats.items())
    assert [list(k) for k, _ in got] == want["patterns"].tolist()
    assert [c for _, c in got] == want["counts"].tolist()
    assert np.allclose(np.concatenate(marg), want["marginals"], rtol=1e-9, atol=1e-15)
    # a second call continues the sample numbering, like the reference's engine keeps its state
    pats2, _ = lw.make_samples(ev, unit, eps)
    want2 = oracle_mod.make_samples(model, ev_state, unit, eps, seed=4242, stream="repo", order=_kahn_min_order(model),
                                    sample_begin=want["units"] * unit)
    assert lw.last_units == want2["units"] and sorted(pats2.items())[0][1] == int(want2["counts"][0])
    lw.engine.close()

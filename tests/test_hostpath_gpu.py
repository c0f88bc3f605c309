"""The drop-in's host path (bn_bp_run / bn_bp_run_view): evidence upload, run and download queued back to back
behind ONE synchronisation must give what the three-step path (set_evidence, run_device, copy_beliefs) gives,
on both execution paths, for changing evidence sets, and when the predicted sweep count is wrong.
Reference: belief_propagation.hpp:31 (one call = evidence in, marginals out), :151-158 (beliefs)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _without_the_dag_path(monkeypatch):
    """These tests are about the host path of the tile kernels; k = 4 networks that fit the chip would by default take the register-resident DAG
    path (bn_dag.hip, tests/test_dag_gpu.py).  BN_DAG sets the option's default for engines created from here on."""
    monkeypatch.setenv("BN_DAG", "0")


@pytest.fixture(scope="module")
def Engine(bnlib):
    from bayesiannetwork_amd.engine import Engine
    return Engine


def _three_steps(eng, ev, eps, max_sweeps=0):
    eng.bp_set_evidence(ev)
    r = eng.bp_run_device(eps, max_sweeps)
    r["beliefs"] = eng.bp_beliefs()
    return r


@pytest.mark.parametrize("multisweep", [0, 2])
def test_run_and_view_equal_three_step_path(Engine, oracle_mod, multisweep):
    from bayesiannetwork_amd import synth
    g = synth.grid(72, 65, 4, seed=21)
    evs = [synth.random_evidence(g, f, seed=3 + q) for q, f in enumerate([0.0, 0.01, 0.2, 0.05, 0.0])]
    with Engine(g) as eng:
        eng.set_option("multisweep", multisweep)
        for ev, eps in zip(evs, [1e-3, 1e-6, 1e-3, 1e-9, 1e-5]):  # sweep counts differ from call to call
            want = oracle_mod.bp_run(g, ev, eps)
            a = eng.bp_run(ev, eps)
            assert eng.last_path() == multisweep
            v = eng.bp_run_view(ev, eps)
            view_copy = v["beliefs"].copy()
            assert np.array_equal(eng.bp_beliefs(), view_copy)   # bn_bp_copy_beliefs after a view: the same run's marginals
            b = _three_steps(eng, ev, eps)
            assert a["sweeps"] == v["sweeps"] == b["sweeps"] == want["sweeps"]
            assert np.array_equal(a["beliefs"], want["beliefs"])
            assert np.array_equal(view_copy, want["beliefs"])
            assert np.array_equal(b["beliefs"], want["beliefs"])
        capped = eng.bp_run_view(evs[1], 1e-12, max_sweeps=3)
        assert capped["sweeps"] == 3
        assert np.array_equal(capped["beliefs"], oracle_mod.bp_run(g, evs[1], 1e-12, max_sweeps=3)["beliefs"])


def test_view_on_lane_group_and_any_arity_tiles(Engine, oracle_mod):
    from bayesiannetwork_amd import synth
    d = synth.random_dag(700, 4, 32, 4, seed=17)
    mx = synth.random_dag(90, 3, 16, [2, 3, 5, 4], seed=8)
    for net in (d, mx):
        with Engine(net) as eng:
            for q in range(3):
                ev = synth.random_evidence(net, 0.03 * q, seed=q)
                want = oracle_mod.bp_run(net, ev, 1e-6)
                got = eng.bp_run_view(ev, 1e-6)
                assert got["sweeps"] == want["sweeps"]
                assert np.abs(got["beliefs"] - want["beliefs"]).max() < 1e-12


def test_stats_report_resident_aborts(Engine):
    from bayesiannetwork_amd import synth
    g = synth.grid(40, 40, 4, seed=2)
    with Engine(g) as eng:
        eng.set_option("multisweep", 2)
        eng.set_option("mid", 0)   # (by default this grid takes the several-workgroup item kernel)
        eng.bp_run(None, 1e-3)
        assert eng.last_path() == 2
        assert eng.bp_stats()["resident_aborts"] == 0


def test_wide_split_batch_equals_single_queries(Engine):
    """lanes_per_node = 3 (latency rules + 16 table entries per lane): a batch runs on the engine's dense twin, which
    must use the SAME lane-group split so that every set keeps the bits of its single run (bn_plan.cpp)."""
    from bayesiannetwork_amd import synth
    g = synth.random_dag(260, 4, 24, 4, seed=23)
    evs = [synth.random_evidence(g, f, seed=q) for q, f in enumerate([0.0, 0.02, 0.1])]
    with Engine(g) as auto, Engine(g, lanes_per_node=2) as dense:   # the automatic layout applies the split by itself ...
        auto.set_option("dag", 0)    # (the tile layouts are what this test is about; by default this network takes bn_dag.hip,
        dense.set_option("dag", 0)   # where a batch is a sequence of single-query launches)
        assert any(c["variant"] == 2 and c["lanes_per_node"] == 4 ** (c["m"] - 1) for c in auto.layout_classes())
        assert all(c["lanes_per_node"] == 4 ** (c["m"] - 2) for c in dense.layout_classes() if c["variant"] == 2)   # ... the dense one never
        a, d = auto.bp_run(evs[1], 1e-6), dense.bp_run(evs[1], 1e-6)
        assert a["sweeps"] == d["sweeps"] and np.abs(a["beliefs"] - d["beliefs"]).max() < 1e-12      # same to rounding
        out = auto.bp_run_batch(evs, 1e-6)
        for q, ev in enumerate(evs):                                                                   # and batches keep the bits
            assert np.array_equal(out["beliefs"][q], auto.bp_run(ev, 1e-6)["beliefs"])
    with Engine(g, lanes_per_node=3) as eng:
        eng.set_option("dag", 0)
        cls = eng.layout_classes()
        # the case must exercise both rules: wide lane groups (m = 3 -> 16 lanes, m = 4 -> 64) and any-arity tiles for
        # one-lane shapes with more than four children
        assert any(c["variant"] == 2 and c["lanes_per_node"] == 4 ** (c["m"] - 1) for c in cls), cls
        assert any(c["variant"] == 3 and c["m"] <= 2 for c in cls), cls
        singles = [eng.bp_run(ev, 1e-6) for ev in evs]
        out = eng.bp_run_batch(evs, 1e-6)
        for q, r in enumerate(singles):
            assert out["sweeps"][q] == r["sweeps"]
            assert np.array_equal(out["beliefs"][q], r["beliefs"]), f"set {q}: batch differs from the single run"


def test_many_evidence_sets_wrap_the_mark_value(Engine, oracle_mod):
    """An evidence set is in force through its mark value (1..255, no clearing between sets); after 255 sets the values
    start over behind a memset.  600 queries alternating between sets that share some nodes and not others."""
    from bayesiannetwork_amd import synth
    g = synth.grid(24, 20, 4, seed=31)
    evs = [synth.random_evidence(g, f, seed=s) for f, s in ((0.1, 1), (0.0, 2), (0.1, 3), (0.3, 1))]
    wants = [oracle_mod.bp_run(g, ev, 1e-6) for ev in evs]
    with Engine(g) as eng:
        for i in range(600):
            q = (i * 7) % 4
            r = eng.bp_run_view(evs[q], 1e-6)
            if i % 37 == 0 or i > 590:
                assert r["sweeps"] == wants[q]["sweeps"] and np.array_equal(r["beliefs"], wants[q]["beliefs"]), i


def test_step_api_after_a_view_run(Engine, oracle_mod):
    """bn_bp_run_view leaves its marginals in the engine's page-locked buffer only; a run through the single-step API afterwards
    writes d_beliefs and the tile buffers: bn_bp_copy_beliefs / bn_bp_beliefs_device / bn_bp_messages must then read THOSE, not the
    view's stale copy or another path's state arrays."""
    from bayesiannetwork_amd import synth
    g = synth.random_dag(400, 3, 16, [2, 3, 4, 3], seed=3)      # the view run takes the item kernel over several workgroups (path 4)
    ev_a = synth.random_evidence(g, 0.05, seed=1)
    ev_b = synth.random_evidence(g, 0.3, seed=2)
    with Engine(g) as eng:
        va = eng.bp_run_view(ev_a, 1e-6)["beliefs"].copy()
        assert eng.last_path() in (3, 4)
        eng.bp_set_evidence(ev_b)
        eng.step_begin()
        o = oracle_mod.bp_run(g, ev_b, 1e-6, dump_msgs=True)
        for s in range(o["sweeps"]):
            eng.step_sweep(s, 1e-6)
        done, sweeps, _ = eng.step_finish(o["sweeps"], False, 1e-6)
        assert done == 1 and sweeps == o["sweeps"] and eng.last_path() == 0
        bel = eng.bp_beliefs()
        assert np.abs(bel - o["beliefs"]).max() < 1e-12 and np.abs(bel - va).max() > 1e-3
        pi, lam = eng.bp_messages()
        assert np.abs(pi - o["pi_msg"]).max() < 1e-12 and np.abs(lam - o["lambda_msg"]).max() < 1e-12


def test_debug_stream_measures_a_plausible_rate(bnlib):
    """bn_debug_stream: the library's own copy / triad kernels reach a few TB/s on an MI355X (bench.py's `hbm_stream_gbs_measured`) and
    check their own output."""
    import ctypes
    for mode in (0, 1):
        g = ctypes.c_double(0.0)
        assert bnlib.bn_debug_stream(-1, mode, 1 << 29, 3, ctypes.byref(g)) == 0
        assert 2000.0 < g.value < 8000.0, (mode, g.value)


def test_construction_split_is_reported(bnlib):
    """bn_get_info "create_us_*" on a device engine: the host plans and the device side add up to (most of) the time bn_create took; a second
    engine of the process -- the runtime is up, a parked stream is at hand -- is created in a few milliseconds."""
    import time
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import Engine
    g = synth.random_dag(30, 4, 12, [2, 3, 4, 3], seed=7)
    with Engine(g) as first:
        first.bp_run(None, 1e-6)
    for _ in range(3):
        t0 = time.perf_counter()
        e = Engine(g)
        dt_ms = (time.perf_counter() - t0) * 1e3
        parts = {k: e.info("create_us_" + k) / 1e3 for k in ("plan", "small", "mid", "dag", "device")}
        e.close()
        assert all(v >= 0 for v in parts.values()) and parts["device"] > 0
        assert sum(parts.values()) <= dt_ms + 0.5
    assert dt_ms < 20.0, (dt_ms, parts)   # (measured: 1.8-2.5 ms; the bound leaves room for a loaded host)

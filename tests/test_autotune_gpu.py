"""Option "autotune": the next run times every execution path the engine is eligible for on the staged evidence and keeps the
fastest (bn_get_info "autotuned_path"); results stay the oracle's.  And what a caller sees when a one-launch path loses the chip:
two functors hammering one device from two threads still answer correctly, aborts (if any) are counted and reported once."""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ELIGIBLE = {2: "resident_eligible", 3: "small_eligible", 4: "mid_eligible", 5: "dag_eligible"}
FORCE = {0: {"multisweep": 0}, 2: {"multisweep": 2, "small": 0, "mid": 0, "dag": 0}, 3: {"small": 2, "mid": 0, "dag": 0},
         4: {"mid": 2, "small": 0, "dag": 0}, 5: {"dag": 2}}
DEFAULTS = {"multisweep": 1, "small": 1, "mid": 1, "dag": 1}


@pytest.fixture(scope="module")
def Engine(bnlib):
    from bayesiannetwork_amd.engine import Engine
    return Engine


def _us_per_query(eng, eps, reps=40):
    for _ in range(5):
        eng.bp_run_device(eps)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.bp_run_device(eps)
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best


@pytest.mark.parametrize("name", ["grid40", "dag1000", "mixed300", "chain200", "grid128"])
def test_autotune_keeps_the_fastest_path(Engine, oracle_mod, name):
    from bayesiannetwork_amd import synth
    g = {"grid40": lambda: synth.grid(40, 40, 4, seed=1), "dag1000": lambda: synth.random_dag(1000, 4, 64, 4, seed=1000),
         "mixed300": lambda: synth.random_dag(300, 3, 32, [2, 3, 4, 3, 2, 5], seed=4), "chain200": lambda: synth.grid(200, 1, 4, seed=5),
         "grid128": lambda: synth.grid(128, 128, 4, seed=1)}[name]()
    ev = synth.random_evidence(g, 0.02, seed=7)
    want = oracle_mod.bp_run(g, ev, 1e-6)
    with Engine(g) as eng:
        assert eng.info("autotuned") == 0
        eng.bp_set_evidence(ev)
        times = {}
        for path, opts in FORCE.items():
            if path in ELIGIBLE and not eng.info(ELIGIBLE[path]):
                continue
            for k, v in {**DEFAULTS, **opts}.items():
                eng.set_option(k, v)
            t = _us_per_query(eng, 1e-6)
            if eng.last_path() == path:
                times[path] = t
        for k, v in DEFAULTS.items():
            eng.set_option(k, v)
        eng.set_option("autotune", 1)
        r = eng.bp_run_device(1e-6)
        assert eng.info("autotuned") == 1
        chosen = eng.info("autotuned_path")
        assert chosen in times and eng.last_path() == chosen
        assert r["sweeps"] == want["sweeps"] and np.abs(eng.bp_beliefs() - want["beliefs"]).max() < 1e-12
        assert times[chosen] <= 1.10 * min(times.values()), (chosen, times)     # within 10 % of the best path measured here
        assert _us_per_query(eng, 1e-6) <= 1.15 * min(times.values())
        eng.set_option("dag", 0)                                               # the options can still be set afterwards
        eng.set_option("multisweep", 0)
        eng.bp_run_device(1e-6)
        assert eng.last_path() == 0


def test_two_functors_on_one_device_from_two_threads(Engine, oracle_mod, capfd):
    """The reference's functors are independent objects (belief_propagation.hpp:320-333); here each holds kernels that want most of
    the chip for themselves.  Two of them used at the same time from two threads: every answer is still the oracle's; when a
    one-launch kernel gives up its bounded wait the run is repeated on a slower path, the event is counted and ONE line per engine
    says so on stderr."""
    from bayesiannetwork_amd import synth
    g = synth.grid(200, 200, 4, seed=1)
    evs = [synth.random_evidence(g, 0.01, seed=30 + q) for q in range(4)]
    wants = [oracle_mod.bp_run(g, ev, 1e-3, threads=8) for ev in evs]
    engines = [Engine(g), Engine(g)]
    errors = []

    def work(eng, offset):
        try:
            for i in range(60):
                q = (i + offset) % 4
                r = eng.bp_run(evs[q], 1e-3)
                if r["sweeps"] != wants[q]["sweeps"] or not np.array_equal(r["beliefs"], wants[q]["beliefs"]):
                    errors.append((offset, i, eng.last_path()))
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))

    threads = [threading.Thread(target=work, args=(e, 2 * j)) for j, e in enumerate(engines)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    aborts = [e.bp_stats()["resident_aborts"] for e in engines]
    paths = [e.last_path() for e in engines]
    for e in engines:
        e.close()
    assert not errors, errors[:5]
    err = capfd.readouterr().err
    if sum(aborts) > 0:
        assert err.count("gave up a bounded wait") == sum(1 for a in aborts if a > 0)   # one line per engine, however many events
    else:
        assert all(p == 2 for p in paths) and "gave up a bounded wait" not in err


@pytest.mark.gpu
def test_default_path_of_small_networks(bnlib, oracle_mod):
    """Which one-launch path a SMALL network takes by default (bn_engine.cpp dag_applies; timings in profiles/r05_paths.json): one or
    two rounds of entry items -> one workgroup, state in LDS (path 3: ALARM-sized); three or more with arities <= 4 and <= 2 parents per
    node -> the register-resident DAG path (path 5: 8 x 8 grid, k = 4, 64 vs 86 us per query); a chain the resident tiles run in one block stays
    there (path 2).  Whatever the path, the oracle's sweep count and marginals."""
    import os
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.dsc import load_dsc
    from bayesiannetwork_amd.engine import Engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    alarm, _ = load_dsc(os.path.join(root, "tests", "golden", "alarm_shaped.dsc"))
    for g, want_path, exact in ((alarm, 3, True), (synth.pearl(), 3, True), (synth.grid(8, 8, 4, seed=1), 5, True),
                                (synth.grid(200, 1, 4, seed=5), 2, True), (synth.random_dag(30, 4, 12, [2, 3, 4, 3], seed=7), 3, True)):
        ev = synth.random_evidence(g, 0.1, seed=3)
        o = oracle_mod.bp_run(g, ev, 1e-6)
        with Engine(g) as eng:
            r = eng.bp_run(ev, 1e-6)
            assert eng.last_path() == want_path, (g.name, eng.last_path())
            assert r["sweeps"] == o["sweeps"]
            assert np.array_equal(r["beliefs"], o["beliefs"]) if exact else np.abs(r["beliefs"] - o["beliefs"]).max() < 1e-12
    # A small network with >= 3-parent nodes stays on the one-workgroup path WHATEVER its rounds of entry items: that path keeps the
    # reference's order for any parent count (the DAG path's lane groups re-associate), and a batch of such a network runs one workgroup
    # per set -- a set's answer must not depend on whether it was asked alone or in a batch (scripts/soak_gpu.py, round 6: a 30-node
    # network of arities {4, 2} with <= 4 parents differed by 2e-16 between the two).
    g = synth.random_dag(30, 4, 8, [4, 2], seed=398548165)
    assert int(np.diff(g.in_ptr).max()) >= 3
    sets = [synth.random_evidence(g, 0.1, seed=3), synth.random_evidence(g, 0.05, seed=5), synth.random_evidence(g, 0.2, seed=6)]
    with Engine(g) as eng:
        assert eng.info("small_eligible") == 1 and eng.info("dag_eligible") == 1
        out = eng.bp_run_batch(sets, 1e-9, 0)
        for q, ev in enumerate(sets):
            o = oracle_mod.bp_run(g, ev, 1e-9)
            r = eng.bp_run(ev, 1e-9)
            assert eng.last_path() == 3 and r["sweeps"] == o["sweeps"] == int(out["sweeps"][q])
            assert np.array_equal(r["beliefs"], o["beliefs"]) and np.array_equal(out["beliefs"][q], r["beliefs"])
        eng.set_option("dag", 2)     # (forced, the DAG path still runs it: to rounding)
        r5 = eng.bp_run(sets[0], 1e-9)
        assert eng.last_path() == 5 and np.abs(r5["beliefs"] - out["beliefs"][0]).max() < 1e-12

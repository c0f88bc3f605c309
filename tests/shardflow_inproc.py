"""n shard engines of one network in ONE process on ONE device, their resident kernels co-resident (one thread per
shard), halo exchange inside the kernels through plain peer pointers -- started by tests/test_shardflow_gpu.py as a
process of its own with GPU_MAX_HW_QUEUES=16: HIP multiplexes a process's streams onto 4 hardware queues by default,
and two persistent kernels that wait for each other must not share one.  (One process per GPU, the deployment shape,
has a queue per rank anyway.)  argv: case names.  Prints INPROC_OK <case> per passed case."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402


def run_threads(shards, eps, max_sweeps):
    out, err = [None] * len(shards), []

    def work(i):
        try:
            out[i] = shards[i].bp_run_device(eps, max_sweeps)
        except Exception as ex:  # noqa: BLE001
            err.append((i, str(ex)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(shards))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    return out


def check(model, ev, eps, nranks, owner=None, max_sweeps=0, reps=3, launches=1, lanes_per_node=0):
    with Engine(model, lanes_per_node=lanes_per_node) as one:
        want = one.bp_run(ev, eps, max_sweeps)
        want_res = one.bp_residuals()
        want_pi, want_lam = one.bp_messages()
    shards = [Engine(model, lanes_per_node=lanes_per_node, rank=r, nranks=nranks, owner=owner) for r in range(nranks)]
    try:
        blobs = [s.peer_export() for s in shards]
        assert all(s.peer_import(blobs) for s in shards), "the in-kernel exchange must be available for this network"
        for s in shards:
            s.bp_set_evidence(ev)
        for rep in range(reps):
            outs = run_threads(shards, eps, max_sweeps)
            for s, o in zip(shards, outs):
                assert s.last_path() == 2 and s.info("last_flow") == 1
                assert s.bp_stats()["resident_aborts"] == 0 and s.bp_stats()["sweep_launches"] == launches
                assert o["sweeps"] == want["sweeps"], f"rep {rep}"
                assert o["residual"] == want["residual"] or (np.isnan(o["residual"]) and np.isnan(want["residual"]))
                assert np.array_equal(s.bp_residuals(), want_res), "every rank must see the same residual history"
            bel = sum(s.bp_beliefs() for s in shards)  # zeros for nodes of other ranks
            assert np.array_equal(bel, want["beliefs"], equal_nan=True), f"rep {rep}"
        moff = model.msg_off
        for s in shards:  # every rank's view of the messages it can see equals the unsharded run
            pi, lam = s.bp_messages()
            rpi, _ = s.edge_refs()
            seen = np.repeat(rpi >= 0, np.diff(moff))
            assert np.array_equal(pi[seen], want_pi[seen], equal_nan=True)
            assert np.array_equal(lam[seen], want_lam[seen], equal_nan=True)
    finally:
        for s in shards:
            s.close()


def grid_case(rows, cols, nranks, eps, k=4, max_sweeps=0, launches=1):
    g = synth.grid(rows, cols, k, seed=rows * 31 + cols)
    check(g, synth.random_evidence(g, 0.02, seed=3), eps, nranks, max_sweeps=max_sweeps, launches=launches)


def worst_cut():
    """A random node -> rank map cuts almost every edge: every tile reports to every rank."""
    g = synth.grid(24, 20, 4, seed=5)
    owner = (synth.splitmix64(9, 0, g.n) % np.uint64(3)).astype(np.int32)
    check(g, synth.random_evidence(g, 0.05, seed=1), 1e-9, 3, owner)


def caps_and_empty_rank():
    g = synth.grid(24, 20, 4, seed=5)
    check(g, None, 1e-12, 2, max_sweeps=5)                       # max_sweeps stops all ranks together
    m = synth.pearl()
    check(m, None, 1e-3, 3, owner=np.array([0, 0, 2, 2], np.int32))  # rank 1 owns nothing: it still exchanges residuals


def tree():
    t = synth.random_dag(900, 2, 8, 4, seed=41)   # <= 2 parents, several children: the all-shapes instantiation
    assert int(np.bincount(t.in_idx, minlength=t.n).max()) <= 8
    # the dense layout keeps nodes with more than 4 children on one-lane tiles (the automatic one moves them to any-arity tiles)
    check(t, synth.random_evidence(t, 0.02, seed=2), 1e-6, 3, lanes_per_node=2)


def mixed_k():
    d = synth.random_dag(700, 2, 8, 3, seed=43)
    if int(np.bincount(d.in_idx, minlength=d.n).max()) <= 8:
        check(d, synth.random_evidence(d, 0.02, seed=2), 1e-6, 2, lanes_per_node=2)


def evidence_changes():
    """A stream of different queries through bn_bp_run (evidence in, marginals out, one call) on every shard: nothing of
    one query may leak into the next (marks take a new value per set, generations count on with the engine's runs)."""
    g = synth.grid(72, 60, 4, seed=13)
    evs = [synth.random_evidence(g, f, seed=5 + q) for q, f in enumerate([0.02, 0.0, 0.1, 0.02, 0.3])]
    with Engine(g) as one:
        wants = [one.bp_run(ev, 1e-6) for ev in evs]
    shards = [Engine(g, rank=r, nranks=3) for r in range(3)]
    try:
        blobs = [s.peer_export() for s in shards]
        assert all(s.peer_import(blobs) for s in shards)
        for ev, want in zip(evs, wants):
            out, err = [None] * 3, []

            def work(i):
                try:
                    out[i] = shards[i].bp_run(ev, 1e-6)
                except Exception as ex:  # noqa: BLE001
                    err.append(str(ex))

            th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert not err, err
            assert all(o["sweeps"] == want["sweeps"] for o in out)
            assert np.array_equal(sum(o["beliefs"] for o in out), want["beliefs"])
            assert all(s.last_path() == 2 and s.bp_stats()["resident_aborts"] == 0 for s in shards)
    finally:
        for s in shards:
            s.close()


CASES = {
    "grid96x80_r2": lambda: grid_case(96, 80, 2, 1e-6),
    "grid96x80_r4": lambda: grid_case(96, 80, 4, 1e-3),
    "grid48x40_r3": lambda: grid_case(48, 40, 3, 1e-9),
    "grid40x33_k3_r2": lambda: grid_case(40, 33, 2, 1e-6, k=3),
    "grid50x50_k2_r3": lambda: grid_case(50, 50, 3, 1e-6, k=2),
    "grid200_r8": lambda: grid_case(200, 200, 8, 1e-3),
    "grid316_r2": lambda: grid_case(316, 316, 2, 1e-3),
    "long_run_two_launches": lambda: grid_case(40, 33, 2, 0.0, k=3, max_sweeps=1100, launches=2),
    "worst_cut": worst_cut,
    "evidence_changes": evidence_changes,
    "caps_and_empty_rank": caps_and_empty_rank,
    "tree": tree,
    "mixed_k": mixed_k,
}

if __name__ == "__main__":
    for name in sys.argv[1:]:
        CASES[name]()
        print("INPROC_OK", name, flush=True)

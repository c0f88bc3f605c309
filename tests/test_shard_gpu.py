"""Multi-GPU data path validated on ONE GPU: n shard engines (the per-rank layouts, kernels with
exchange-resident cut edges, residual slots in the segments, stopping logic) are driven through
the step API with an emulated all-gather (bn_debug_allgather).  Only the RCCL call itself is not
exercised here.  Results must be BIT-IDENTICAL to the unsharded engine: Jacobi sweeps do not depend
on who computes a message."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(bnlib):
    from bayesiannetwork_amd import engine
    return engine


def _run_sharded(eng, model, ev, eps, nranks, owner=None, max_sweeps=0, overlapped=True):
    shards = [eng.Engine(model, rank=r, nranks=nranks, owner=owner) for r in range(nranks)]
    try:
        out = eng.run_shards_on_one_device(shards, ev, eps, max_sweeps, overlapped=overlapped)
        bel = sum(s.bp_beliefs() for s in shards)  # zeros for nodes of other ranks
        res = shards[0].bp_residuals()
        for s in shards[1:]:
            assert np.array_equal(s.bp_residuals(), res), "every rank must see the same residual history"
        msgs = [s.bp_messages() for s in shards]
        refs = [s.edge_refs() for s in shards]
    finally:
        for s in shards:
            s.close()
    return out, bel, res, msgs, refs


def _check(eng, model, ev, eps, nranks, owner=None):
    with eng.Engine(model) as single:
        single.set_option("small", 0)   # the shards run the tile kernels: the reference run too (same bits)
        single.set_option("mid", 0)
        single.set_option("dag", 0)
        want = single.bp_run(ev, eps)
        want_res = single.bp_residuals()
        want_pi, want_lam = single.bp_messages()
    # the launch order of the overlapped run (interior tiles before the previous exchange lands) and the
    # plain order (whole sweep, then exchange) must both reproduce the unsharded run
    out_plain = _run_sharded(eng, model, ev, eps, nranks, owner, overlapped=False)
    out, bel, res, msgs, refs = _run_sharded(eng, model, ev, eps, nranks, owner)
    assert out_plain[0]["sweeps"] == want["sweeps"] and np.array_equal(out_plain[1], want["beliefs"], equal_nan=True)
    assert out["sweeps"] == want["sweeps"]
    assert np.array_equal(res, want_res)
    assert np.array_equal(bel, want["beliefs"], equal_nan=True)
    # every rank's view of the messages it can see equals the unsharded run
    moff = model.msg_off
    for (pi, lam), (rpi, _) in zip(msgs, refs):
        seen = np.repeat(rpi >= 0, np.diff(moff))
        assert np.array_equal(pi[seen], want_pi[seen], equal_nan=True)
        assert np.array_equal(lam[seen], want_lam[seen], equal_nan=True)


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_sharded_grid(eng, nranks):
    from bayesiannetwork_amd import synth
    g = synth.grid(48, 40, 4, seed=11)
    _check(eng, g, synth.random_evidence(g, 0.03, seed=2), 1e-6, nranks)


@pytest.mark.parametrize("nranks,maxp,k", [(2, 2, 4), (4, 4, 4), (3, 3, [2, 3, 4])])
def test_sharded_dag(eng, nranks, maxp, k):
    from bayesiannetwork_amd import synth
    d = synth.random_dag(900, maxp, 48, k, seed=77)
    _check(eng, d, synth.random_evidence(d, 0.02, seed=3), 1e-6, nranks)


def test_sharded_random_owner_worst_case_cut(eng):
    """A random node->rank map cuts almost every edge: every tile is a boundary tile."""
    from bayesiannetwork_amd import synth
    g = synth.grid(20, 20, 4, seed=5)
    owner = (synth.splitmix64(9, 0, g.n) % np.uint64(4)).astype(np.int32)
    _check(eng, g, synth.random_evidence(g, 0.05, seed=1), 1e-9, 4, owner)
    d = synth.random_dag(300, 4, 32, [3, 2, 4], seed=6)
    owner = (synth.splitmix64(10, 0, d.n) % np.uint64(3)).astype(np.int32)
    _check(eng, d, synth.random_evidence(d, 0.05, seed=1), 1e-6, 3, owner)


def test_sharded_empty_rank_and_cap(eng):
    """More ranks than a tiny graph can fill; and max_sweeps stops all ranks together."""
    from bayesiannetwork_amd import synth
    m = synth.pearl()
    _check(eng, m, None, 1e-3, 3, owner=np.array([0, 0, 2, 2], np.int32))  # rank 1 owns nothing
    g = synth.grid(16, 16, 4, seed=3)
    out, bel, _, _, _ = _run_sharded(eng, g, None, 1e-12, 2, max_sweeps=5)
    with eng.Engine(g) as single:
        want = single.bp_run(None, 1e-12, max_sweeps=5)
    assert out["sweeps"] == 5 == want["sweeps"]
    assert np.array_equal(bel, want["beliefs"])


def test_sharded_full_size_config4(eng):
    """BASELINE config 4: the 316x316 grid in 8 row stripes (emulated exchange), vs one engine."""
    from bayesiannetwork_amd import synth
    g = synth.grid(316, 316, 4, seed=2)
    _check(eng, g, synth.random_evidence(g, 0.01, seed=7), 1e-3, 8)


def test_rccl_single_rank_communicator(eng, monkeypatch):
    """RCCL is loadable, a 1-rank communicator initialises, and the per-sweep ncclAllGather itself runs
    (BN_EXCHANGE_ALWAYS makes the launch path issue the -- then trivial -- collective on one rank; the
    n-rank path cannot run on a one-GPU box)."""
    from bayesiannetwork_amd import synth
    uid = eng.Engine.comm_unique_id()
    assert len(uid) == 128
    g = synth.grid(24, 24, 4, seed=1)
    with eng.Engine(g) as e:
        want = e.bp_run(None, 1e-6)
        e.comm_init(uid)
        monkeypatch.setenv("BN_EXCHANGE_ALWAYS", "1")
        e.set_option("multisweep", 0)  # the one-launch paths have no exchange step
        r = e.bp_run(None, 1e-6)
        assert e.last_path() == 0
        assert r["sweeps"] == want["sweeps"] and np.array_equal(r["beliefs"], want["beliefs"])


def test_bench_multi_gpu_code_path_on_one_rank():
    """bench.py's N > 1 branch (shard engine, RCCL communicator, control plane, weak-scaling and
    replicated-queries extras) run end to end as a 1-rank world: the only way to execute that code
    on a one-GPU box.  A child process: the bench initialises torch.distributed."""
    import json
    import os
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:  # a port nobody is listening on
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, BN_FORCE_MULTI="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--rows", "48", "--cols", "48"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    from helpers import parse_bench_output
    assert p.stdout.strip().splitlines()[-1].startswith('{"metric"'), p.stdout[-1500:]
    line, extras = parse_bench_output(p.stdout)   # the LAST stdout line is the contract line
    assert len(p.stdout.strip().splitlines()[-1].encode()) <= 4096
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["unit"] == "edge-messages/s"
    assert extras["weak_scaling"]["value"] > 0
    assert extras["replicated_queries"]["value"] > 0
    for k in ("metric", "steps", "warmup", "ms_per_step", "scaling", "roofline", "config"):
        assert k in line
    # what the communicator itself reports, and the exchange in force
    assert line["config"]["rccl_ranks"] == 1 and line["config"]["world_size"] == 1
    assert line["config"]["exchange"] in ("rccl all-gather per sweep", "in-kernel (peer-mapped memory)")


def test_bench_default_line_is_the_contract_and_small():
    """`python bench.py --gpus 1` as the driver types it (reduced steps, no CPU legs, headline only): the last stdout line is
    the contract line -- every contract key, scalar roofline with frac and frac_survey_8d, config.workload = configs[2],
    dtype f64 -- in at most 4 KB, with the full records on the lines before it."""
    import os
    import subprocess
    import sys
    from bayesiannetwork_amd import benchline
    from helpers import parse_bench_output
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BN_FORCE_MULTI")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    last = p.stdout.strip().splitlines()[-1]
    assert len(last.encode()) <= benchline.MAX_LINE_BYTES
    line, extras = parse_bench_output(p.stdout)
    assert list(line.keys()) == [k for k in benchline.CONTRACT_KEYS if k in line]
    for k in benchline.CONTRACT_KEYS:
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["warmup"] == 2 and line["dtype"] == "f64" and line["vs_baseline"] is None
    assert "configs[2]" in line["config"]["workload"] and line["value"] > 1e9
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "valu") and 0 < roof["frac"] <= 1.0 and roof["frac_survey_8d"] > 0 and roof["kernel"] == "bp_resident_kernel"
    assert roof["avg_launch_us"] > 0 and roof["sweeps_per_launch"] >= 1 and roof["hbm_stream_gbs_measured"] > 1000
    assert all(not isinstance(v, (dict, list)) for v in roof.values())
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["value"] > 0
    assert "roofline_full" in extras

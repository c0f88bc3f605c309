"""Halo exchange INSIDE the resident kernel (bn_peer_export / bn_peer_import): the shards of a network run one
launch each and talk through peer-mapped memory -- cut-edge message halves stored into the peer's exchange
region, per-tile generation granules, per-rank residual granules -- instead of one RCCL all-gather per sweep.
Exercised on ONE device: (a) n shard engines in one process, their kernels co-resident, one thread per shard;
(b) n processes sharing device 0, the peers' buffers mapped through hipIpc handles.  Jacobi sweeps do not depend
on who computes a message (belief_propagation.hpp:78-101, :135-143): beliefs, sweep count and residual history
must be BIT-IDENTICAL to the unsharded engine's."""
import os
import subprocess
import sys
import tempfile
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng(bnlib):
    from bayesiannetwork_amd import engine
    return engine


def _single(eng, model, ev, eps, max_sweeps=0):
    with eng.Engine(model) as one:
        want = one.bp_run(ev, eps, max_sweeps)
        return want, one.bp_residuals(), one.bp_messages()


@pytest.mark.parametrize("cases,waves", [
    (("grid96x80_r2", "grid96x80_r4", "grid48x40_r3", "grid40x33_k3_r2", "grid50x50_k2_r3"), None),
    (("grid200_r8",), None),
    (("grid316_r2",), "8"),
    (("worst_cut", "caps_and_empty_rank", "tree", "mixed_k", "long_run_two_launches", "evidence_changes"), None),
    (("grid96x80_r2", "tree", "evidence_changes"), "8")])
def test_shards_in_one_process(cases, waves):
    """2-8 shard engines in ONE process, kernels co-resident on one device (tests/shardflow_inproc.py; a process of its
    own because the number of hardware queues per process is fixed when HIP starts).  waves None: what the engine picks --
    four waves per block, one per SIMD, with the whole register file, on shards this small (the instantiations a shard of
    the 316x316 grid runs on a GPU of its own); "8": eight waves per block, which the two halves of the 316x316 grid need
    to be co-resident on ONE chip."""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    if waves:
        env["BN_RESIDENT_WAVES"] = waves
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shardflow_inproc.py"), *cases], env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    for c in cases:
        assert f"INPROC_OK {c}" in p.stdout


def test_networks_outside_the_resident_kernel_stay_on_rccl(eng):
    from bayesiannetwork_amd import synth
    d = synth.random_dag(600, 4, 32, 4, seed=7)   # 3-4 parents: lane-group tiles
    shards = [eng.Engine(d, rank=r, nranks=2) for r in range(2)]
    try:
        blobs = [s.peer_export() for s in shards]
        assert not any(s.peer_import(blobs) for s in shards)
    finally:
        for s in shards:
            s.close()


@pytest.mark.parametrize("rows,cols,nranks,eps,max_sweeps", [(316, 316, 2, 1e-3, 0), (120, 100, 4, 1e-6, 0), (64, 64, 2, 0.0, 1100)])
def test_shards_in_separate_processes_over_ipc(eng, rows, cols, nranks, eps, max_sweeps):
    """world_size n on ONE GPU: n processes, each owning a shard on device 0, peers mapped with hipIpcOpenMemHandle --
    the arrangement of a real node (one process per GPU) minus the xGMI links.  (64x64, eps 0, 1100 sweeps: a run that
    takes two launches per rank.)"""
    from bayesiannetwork_amd import synth
    g = synth.grid(rows, cols, 4, seed=rows * 31 + cols)
    ev = synth.random_evidence(g, 0.02, seed=3)
    want, want_res, _ = _single(eng, g, ev, eps, max_sweeps)
    reps = 3
    with tempfile.TemporaryDirectory() as work:
        env = dict(os.environ)
        if rows * cols > 40000:
            env["BN_RESIDENT_WAVES"] = "8"  # the ranks share ONE chip here: keep their blocks few enough to be co-resident
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shardflow_worker.py"), str(r), str(nranks), work,
                                   str(rows), str(cols), repr(eps), str(max_sweeps), str(reps), "0"], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(nranks)]
        logs = []
        for p in procs:
            try:
                logs.append(p.communicate(timeout=600)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()   # exactly the processes started here
                raise
        assert all(p.returncode == 0 for p in procs), "\n".join(l[-1500:] for l in logs)
        outs = [np.load(os.path.join(work, f"out{r}.npz")) for r in range(nranks)]
    for o in outs:
        assert bool(o["shard_flow"]) and int(o["aborts"]) == 0
    for i in range(reps):
        for o in outs:
            assert int(o[f"path{i}"]) == 2 and int(o[f"flow{i}"]) == 1
            assert int(o[f"sweeps{i}"]) == want["sweeps"]
            assert float(o[f"residual{i}"]) == want["residual"]
            assert np.array_equal(o[f"history{i}"], want_res)
        bel = sum(o[f"beliefs{i}"] for o in outs)
        assert np.array_equal(bel, want["beliefs"], equal_nan=True), f"rep {i}"


def test_bench_two_ranks_on_one_device():
    """bench.py --gpus 2 as two torchrun processes that share device 0 (BN_BENCH_SAME_DEVICE; no RCCL communicator can
    span two ranks on one GPU, so BN_NO_RCCL): the N > 1 branch with the in-kernel exchange end to end -- blobs through
    torch.distributed, verification against the unsharded run, timed runs, weak-scaling and replicated-queries legs."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, BN_BENCH_SAME_DEVICE="1", BN_NO_RCCL="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--rows", "96", "--cols", "80"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    from helpers import parse_bench_output
    line, extras = parse_bench_output(p.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert line["config"]["exchange"].startswith("in-kernel"), line["config"]
    assert extras["weak_scaling"]["exchange"].startswith("in-kernel") and extras["weak_scaling"]["value"] > 0
    assert extras["replicated_queries"]["value"] > 0


def test_bench_gpus_2_typed_plainly():
    """`python bench.py --gpus 2` with no launcher around it (the form of the driver's N = 1 command): the parent starts the two
    ranks itself as fresh processes (benchline.launch_ranks), relays rank 0's line as its own last line and exits 0.  Both ranks
    on device 0 (BN_BENCH_SAME_DEVICE, BN_NO_RCCL): the only arrangement a one-GPU box allows."""
    from helpers import parse_bench_output
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BN_BENCH_SAME_DEVICE="1", BN_NO_RCCL="1", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--rows", "96",
                        "--cols", "80", "--no-weak", "--no-replicas"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert len(p.stdout.strip().splitlines()[-1].encode()) <= 4096
    line, _ = parse_bench_output(p.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["steps"] == 5 and line["warmup"] == 2
    assert line["config"]["world_size"] == 2 and line["config"]["rccl_ranks"] == 0   # (BN_NO_RCCL: no communicator was created)
    assert line["config"]["exchange"].startswith("in-kernel")


def test_bench_gpus_4_typed_plainly():
    """... and with four ranks (three stripe cuts, every rank both a producer and a consumer of two neighbours): the N-rank blobs, the
    verification against the unsharded run on every rank, the timed runs and the extras, all on device 0."""
    from helpers import parse_bench_output
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BN_BENCH_SAME_DEVICE="1", BN_NO_RCCL="1", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "4", "--warmup", "2", "--rows", "128",
                        "--cols", "80"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line, extras = parse_bench_output(p.stdout)
    assert line["n_gpus"] == 4 and line["value"] > 0 and line["config"]["world_size"] == 4
    assert line["config"]["exchange"].startswith("in-kernel") and line["config"]["in_kernel_exchange_verified"] is True
    assert extras["replicated_queries"]["value"] > 0

"""N>1 path on CPU: world_size-2 (and 3) gloo jobs, see tests/dist_worker.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case,world,order", [("grid", 2, "plain"), ("dag", 2, "plain"), ("dag", 3, "plain"),
                                              ("grid", 2, "overlapped"), ("dag", 3, "overlapped"),
                                              ("grid", 2, "granules"), ("dag", 3, "granules")])
def test_gloo_sharded_exchange(bnlib, oracle_mod, case, world, order, tmp_path):
    """order "overlapped": the engine's overlapped schedule -- interior tiles of the next sweep computed
    while the all-gather is in flight, cut-touching tiles after it landed; "granules": the in-kernel exchange of
    the sharded resident kernel -- halves pushed to the one rank that reads them, per-tile generation granules
    checked against the neighbour tables bn_peer_import builds, per-rank residuals (see dist_worker.py)."""
    import socket
    with socket.socket() as sk:  # a port nobody is listening on (a fixed one may still be in TIME_WAIT from another run)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), case, order]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert f"DIST_OK case={case} world={world} order={order}" in p.stdout


@pytest.mark.parametrize("failing", ["none", "0", "1"])
def test_run_collective_resolves_a_failed_in_kernel_run_on_every_rank(failing):
    """multigpu.run_collective (what a sharded caller runs a query through): a rank whose in-kernel exchange gave up gets
    BN_ERR_STATE from the library, which does not fall back on its own; all ranks then switch to the RCCL exchange together and
    repeat the run -- also the rank whose first run had succeeded.  World 2 over gloo, a stand-in engine (tests/collective_worker.py)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "collective_worker.py"), failing]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"), cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert f"COLLECTIVE_OK failing={failing} world=2" in p.stdout


@pytest.mark.parametrize("failing,mode,fail_at", [("none", "ok", 0), ("1", "raise", 2), ("0", "raise", 7), ("1", "bits", 5), ("0", "bits", 12)])
def test_verify_in_kernel_exchange_keeps_one_collective_schedule(failing, mode, fail_at):
    """multigpu.verify_in_kernel_exchange (what bench.py --gpus N runs before it times the in-kernel exchange): when one rank's
    verification run raises or returns other bits while its peer's succeeds, both ranks leave the loop after the same run with the
    same answer and the same number of collectives issued (a barrier before and an all-reduce after every run) -- no hang, no
    off-by-one collective.  World 2 over gloo, stand-in engines (tests/verify_worker.py)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "verify_worker.py"), failing, mode, str(fail_at)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=dict(os.environ, OMP_NUM_THREADS="1"), cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert f"VERIFY_OK failing={failing} mode={mode} world=2" in p.stdout

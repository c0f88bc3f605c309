"""N>1 path on CPU: world_size-2 (and 3) gloo jobs, see tests/dist_worker.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case,world", [("grid", 2), ("dag", 2), ("dag", 3)])
def test_gloo_sharded_exchange(bnlib, oracle_mod, case, world, tmp_path):
    port = 29600 + world * 7 + (0 if case == "grid" else 3)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), case]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert f"DIST_OK case={case} world={world}" in p.stdout

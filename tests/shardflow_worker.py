"""One rank of a sharded run with the halo exchange inside the resident kernel, as a process of its own
(tests/test_shardflow_gpu.py starts several of them on ONE device: the peers' buffers arrive as hipIpc handles).
argv: rank nranks workdir rows cols eps max_sweeps reps device"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402


def wait_for(paths, timeout=120.0):
    t0 = time.time()
    while not all(os.path.exists(p) for p in paths):
        if time.time() - t0 > timeout:
            raise SystemExit(f"timeout waiting for {paths}")
        time.sleep(0.01)


def main():
    rank, nranks, work = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    rows, cols, eps, max_sweeps, reps, device = int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), int(sys.argv[9])
    g = synth.grid(rows, cols, 4, seed=rows * 31 + cols)
    ev = synth.random_evidence(g, 0.02, seed=3)
    eng = Engine(g, device=device, rank=rank, nranks=nranks)
    blob = eng.peer_export()
    with open(os.path.join(work, f"blob{rank}.tmp"), "wb") as f:
        f.write(blob)
    os.rename(os.path.join(work, f"blob{rank}.tmp"), os.path.join(work, f"blob{rank}"))
    wait_for([os.path.join(work, f"blob{r}") for r in range(nranks)])
    ok = eng.peer_import([open(os.path.join(work, f"blob{r}"), "rb").read() for r in range(nranks)])
    eng.bp_set_evidence(ev)
    out = {"shard_flow": ok}
    for i in range(reps):
        # every rank enters a run at about the same time (the kernels wait for each other, bounded)
        open(os.path.join(work, f"ready{i}_{rank}"), "w").close()
        wait_for([os.path.join(work, f"ready{i}_{r}") for r in range(nranks)])
        r = eng.bp_run_device(eps, max_sweeps)
        out[f"sweeps{i}"] = r["sweeps"]
        out[f"residual{i}"] = r["residual"]
        out[f"beliefs{i}"] = eng.bp_beliefs()
        out[f"history{i}"] = eng.bp_residuals()
        out[f"path{i}"] = eng.last_path()
        out[f"flow{i}"] = eng.info("last_flow")
    out["aborts"] = eng.bp_stats()["resident_aborts"]
    np.savez(os.path.join(work, f"out{rank}.npz"), **out)
    # nobody unmaps while a peer may still be running
    open(os.path.join(work, f"done_{rank}"), "w").close()
    wait_for([os.path.join(work, f"done_{r}") for r in range(nranks)])
    eng.close()


if __name__ == "__main__":
    main()

# torchrun --nproc-per-node 2 scripts/experiments/two_rank_probe.py [one] [cuda]   (both ranks on device 0)
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch, torch.distributed as dist
from bayesiannetwork_amd import synth
from bayesiannetwork_amd.engine import Engine
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
flags = sys.argv[1:]
if "cuda" in flags:
    torch.cuda.set_device(0); torch.cuda.synchronize()
rows = 316
g = synth.grid(rows, rows, 4, seed=2)
ev = synth.random_evidence(g, 0.01, seed=7)
eng = Engine(g, device=0, rank=rank, nranks=world)
blobs = [None] * world
dist.all_gather_object(blobs, eng.peer_export())
print(rank, "import", eng.peer_import(blobs), "tiles", eng.layout()["n_tiles"], "blocks", eng.info("resident_blocks"), flush=True)
eng.bp_set_evidence(ev)
if "one" in flags:
    with Engine(g, device=0) as one:
        want = one.bp_run(ev, 1e-3)
    print(rank, "unsharded sweeps", want["sweeps"], flush=True)
for i in range(4):
    dist.barrier()
    t0 = time.perf_counter()
    try:
        r = eng.bp_run_device(1e-3)
        print(rank, "run", i, r["sweeps"], "path", eng.last_path(), "aborts", eng.bp_stats()["resident_aborts"], f"{(time.perf_counter()-t0)*1e6:.0f} us", "devclock/sweep", eng.bp_stats()["sweep_devclock_ms"]*1e3/max(r["sweeps"],1), flush=True)
    except Exception as ex:
        print(rank, "run", i, "FAILED", str(ex)[:120], f"{(time.perf_counter()-t0)*1e3:.0f} ms", flush=True)
dist.barrier()
eng.close()

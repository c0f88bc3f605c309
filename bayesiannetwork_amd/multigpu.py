"""One process per GPU: shard engines + the library's RCCL communicator, bootstrapped through
torch.distributed (which is used only as the control plane: unique-id broadcast, barriers,
max-over-ranks timing).  The data path -- one in-place all-gather of the exchange segments per
sweep -- runs inside libbn_mi355x.so over RCCL/xGMI."""
from __future__ import annotations

import datetime
import json
import os
import time

import numpy as np

from . import benchline, synth
from .engine import Engine


def init_control_plane(backend: str = "gloo"):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")        # a lone process (BN_FORCE_MULTI, tests) is a 1-rank world
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300))
    return dist


def make_shard(model, rank: int, world: int, device: int, owner=None, in_kernel: bool = True) -> Engine:
    """Create shard `rank`, join the library's RCCL communicator (the per-sweep all-gather path, also the fallback) and --
    in_kernel -- set up the halo exchange inside the resident kernel: every rank's blob (bn_peer_export: hipIpc handles of
    its record buffers and sync block, tiles of its nodes on cut edges) goes to every rank (bn_peer_import).
    Collective: every rank calls it."""
    dist = init_control_plane()
    eng = Engine(model, device=device, rank=rank, nranks=world, owner=owner)
    if not os.environ.get("BN_NO_RCCL"):  # (several ranks on ONE device, a test arrangement, cannot form an RCCL communicator)
        box = [Engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        eng.comm_init(box[0])
    if in_kernel and world > 1:
        import torch
        blobs = [None] * world
        dist.all_gather_object(blobs, eng.peer_export())
        ok = 1
        try:
            eng.peer_import(blobs)
        except Exception as ex:  # noqa: BLE001 - e.g. a peer's buffers cannot be mapped on this node
            print(f"[multigpu] rank {rank}: bn_peer_import failed ({ex}); staying on per-sweep launches + RCCL", flush=True)
            ok = 0
        t = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)  # every rank or none
        if int(t[0]) == 0:
            eng.set_option("multisweep", 0)
    return eng


def verify_in_kernel_exchange(eng: Engine, model, ev, eps: float, device: int, repeats: int = 3) -> bool:
    """The in-kernel exchange relies on system-scope stores into peer-mapped fine-grained memory being visible to the
    peer's polls in order; before anything is timed, one run of it is compared, on this node's links, with the UNSHARDED
    run of the same query on this rank's own GPU: same sweep count, same residual history, same bits in the beliefs of
    the nodes this rank owns.  On any difference -- on ANY rank -- every rank switches to per-sweep launches + RCCL.
    Collective."""
    import torch
    dist = init_control_plane()
    have = torch.tensor([1 if eng.info("shard_flow") else 0], dtype=torch.int32)
    dist.all_reduce(have, op=dist.ReduceOp.MIN)  # the same answer, and the same collectives below, on every rank
    if int(have[0]) == 0:
        eng.set_option("multisweep", 0)
        return False
    ok = 1
    # several evidence sets, each run `repeats` times: an ordering violation across the links would be intermittent, one clean
    # run does not exclude it.  The last set is the caller's, left staged.
    sets = [synth.random_evidence(model, 0.02, seed=900 + q) for q in range(3)] + [ev]
    owned = np.repeat(eng.node_slots() >= 0, model.k)
    with Engine(model, device=device) as one:
        wants = [(one.bp_run(e_, eps), one.bp_residuals()) for e_ in sets]
    # Every rank executes the SAME sequence of collectives whatever happens to its own runs: one barrier before and one
    # all-reduce(MIN) of the verdict after every run, and all ranks leave the loop together on the first failure anywhere.  (A rank
    # that jumped out on an exception while its peers went on to their next barrier would pair a barrier with an all-reduce.)
    flag = torch.zeros(1, dtype=torch.int32)
    for e_, (want, want_hist) in zip(sets, wants):
        try:
            eng.bp_set_evidence(e_)
        except Exception as ex:  # noqa: BLE001
            print(f"[multigpu] rank {eng.rank}: staging evidence for the verification failed: {ex}", flush=True)
            ok = 0
        for _ in range(repeats):
            dist.barrier()  # the ranks' kernels wait for each other (bounded: 2 s): enter the run together
            if ok:
                try:
                    got = eng.bp_run_device(eps)
                    bel = eng.bp_beliefs()
                    if (eng.last_path() != 2 or got["sweeps"] != want["sweeps"] or not np.array_equal(eng.bp_residuals(), want_hist)
                            or not np.array_equal(bel[owned], want["beliefs"][owned], equal_nan=True)):
                        ok = 0
                except Exception as ex:  # noqa: BLE001 - e.g. a bounded wait gave up
                    print(f"[multigpu] rank {eng.rank}: in-kernel exchange failed verification: {ex}", flush=True)
                    ok = 0
            flag[0] = ok
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag[0])
            if not ok:
                break
        if not ok:
            break
    if not ok:
        eng.set_option("multisweep", 0)
        try:
            eng.bp_set_evidence(ev)   # the caller's set stays staged, as after a clean verification
        except Exception:  # noqa: BLE001
            pass
        return False
    return True


def run_collective(eng: Engine, eps: float, max_sweeps: int = 0):
    """One sharded run with the outcome agreed by ALL ranks.  A rank whose in-kernel exchange gives up a bounded wait gets
    BN_ERR_STATE from the library, which does NOT fall back by itself (a peer may have finished the same run successfully and would
    never join the RCCL collective of a unilateral fall-back).  Here every rank reports, the minimum is all-reduced over the
    control plane, and on any failure EVERY rank switches to per-sweep launches + one RCCL all-gather per sweep ("multisweep" 0)
    and repeats the run.  Collective: every rank calls it for every run.  Returns (result, in_kernel_still_on)."""
    import torch
    from . import _lib
    dist = init_control_plane()
    ok, res = 1, None
    try:
        res = eng.bp_run_device(eps, max_sweeps)
    except _lib.BnError as ex:
        if ex.code != _lib.BN_ERR_STATE:
            raise
        print(f"[multigpu] rank {eng.rank}: {ex}", flush=True)
        ok = 0
    t = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if int(t[0]) == 1:
        return res, True
    eng.set_option("multisweep", 0)
    dist.barrier()                       # every rank's kernel of the failed run has ended before any rank's fall-back run starts
    return eng.bp_run_device(eps, max_sweeps), False


def gather_beliefs(eng: Engine) -> np.ndarray:
    """Global node-major beliefs on every rank: shards hold zeros for nodes they do not own."""
    import torch
    dist = init_control_plane()
    t = torch.from_numpy(eng.bp_beliefs())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()


def _timed_runs(eng, eps, steps, warmup, dist, torch):
    """-> (seconds, sweeps, device-clock ms, launches, ok).  ok False: some rank's in-kernel exchange gave up a bounded wait
    (BN_ERR_STATE; the library does not fall back by itself on sharded engines) -- agreed over all ranks AFTER the loop, so that
    no collective sits inside the timed region; the caller then switches every rank to the RCCL exchange and times again."""
    from . import _lib
    failed = 0

    def run():
        nonlocal failed
        if failed:
            return None
        try:
            return eng.bp_run_device(eps)
        except _lib.BnError as ex:
            if ex.code != _lib.BN_ERR_STATE:
                raise
            print(f"[multigpu] rank {eng.rank}: {ex}", flush=True)
            failed = 1
            return None

    for _ in range(max(warmup, 1)):
        run()
    torch.cuda.synchronize()
    dist.barrier()
    sweeps, kern_ms, launches = 0, 0.0, 0
    t0 = time.perf_counter()
    for _ in range(steps):
        r = run()
        if r is None:
            continue
        st = eng.bp_stats()
        sweeps += r["sweeps"]
        kern_ms += st["sweep_devclock_ms"]  # device clock: first sweep's start -> last sweep's end, exchanges included
        launches += r["sweeps"]
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    f = torch.tensor([failed], dtype=torch.int32)
    dist.all_reduce(f, op=dist.ReduceOp.MAX)
    return float(dt[0]), sweeps, kern_ms, launches, int(f[0]) == 0


def bench_main(a, rank: int, world: int, local_rank: int) -> None:
    """bench.py --gpus N (N > 1): BASELINE.json configs[3] -- the SAME 316x316 grid cut into N row
    stripes (strong scaling) -- plus, as an extra key, the weak-scaling variant (316 rows per GPU)."""
    import threading
    import torch
    # a collective that never completes (a rank lost, RCCL unable to bring up a link) must not hang
    # the box: give up loudly after BN_BENCH_WATCHDOG_S seconds
    limit = float(os.environ.get("BN_BENCH_WATCHDOG_S", "600"))

    def _give_up():
        print(f"[bench] rank {rank}: multi-GPU bench exceeded {limit:.0f} s -- aborting", flush=True)
        os._exit(124)

    dog = threading.Timer(limit, _give_up)
    dog.daemon = True
    dog.start()
    dist = init_control_plane()
    # BN_BENCH_SAME_DEVICE (with BN_NO_RCCL): every rank on device 0 -- the N-process arrangement on a one-GPU box (tests)
    device = 0 if os.environ.get("BN_BENCH_SAME_DEVICE") else local_rank
    torch.cuda.set_device(device)
    have_rccl = not os.environ.get("BN_NO_RCCL")
    out = None
    g = synth.grid(a.rows, a.cols, 4, seed=2)
    ev = synth.random_evidence(g, a.evidence, seed=7)
    eng = make_shard(g, rank, world, device, in_kernel=not os.environ.get("BN_NO_PEER_EXCHANGE"))
    eng.bp_set_evidence(ev)
    verified = verify_in_kernel_exchange(eng, g, ev, a.eps, device)
    # Which exchange `value` reports.  The in-kernel exchange rests on system-scope write-through stores into peer memory becoming
    # visible in drain order; until it has run on an xGMI node (no multi-GPU node was available to the builder) it is the headline
    # only where it is asked for (BN_PEER_EXCHANGE=1) or where every rank shares one device (the test arrangement); on a real
    # multi-device world the RCCL all-gather is, with the verified in-kernel figure beside it.
    peer_headline = bool(os.environ.get("BN_PEER_EXCHANGE") == "1" or os.environ.get("BN_BENCH_SAME_DEVICE") or not have_rccl)
    in_kernel = verified and peer_headline
    if not verified and not have_rccl:
        raise SystemExit("in-kernel exchange not available and BN_NO_RCCL set: nothing to run the shards with")
    eng.set_option("multisweep", 1 if in_kernel else 0)
    dt, sweeps, kern_ms, launches, ok = _timed_runs(eng, a.eps, a.steps, a.warmup, dist, torch)
    if not ok:  # some rank's in-kernel exchange gave up: every rank to the RCCL exchange, timed again
        if not have_rccl:
            raise SystemExit("the in-kernel exchange gave up a bounded wait and BN_NO_RCCL is set")
        in_kernel = verified = False
        eng.set_option("multisweep", 0)
        dt, sweeps, kern_ms, launches, ok = _timed_runs(eng, a.eps, a.steps, a.warmup, dist, torch)
    path = eng.last_path()
    rccl = None
    other = None
    if verified and have_rccl:  # the same shards through the other exchange, for comparison
        eng.set_option("multisweep", 0 if in_kernel else 1)
        dtr, sr, _, _, ok2 = _timed_runs(eng, a.eps, max(a.steps // 2, 3), 2, dist, torch)
        eng.set_option("multisweep", 1 if in_kernel else 0)
        fig = {"value": g.messages_per_sweep() * sr / dtr if ok2 else None, "ms_per_step": dtr / max(a.steps // 2, 3) * 1e3}
        if in_kernel:
            rccl = dict(fig, what="per-sweep launches + one in-place RCCL all-gather per sweep, overlapped with the interior tiles")
        else:
            other = dict(fig, what="halo exchange inside the resident kernel (peer-mapped memory), verified on this node against the unsharded "
                                   "run over 4 evidence sets x 3 runs before timing; BN_PEER_EXCHANGE=1 makes it the headline")
    li = eng.layout()
    seg = li["segment_bytes"]
    rccl_ranks = int(eng.info("rccl_ranks"))
    if have_rccl and rccl_ranks not in (0, world):   # (0: the library's librccl has no ncclCommCount -- nothing to compare)
        raise SystemExit(f"rank {rank}: the RCCL communicator reports {rccl_ranks} ranks in a world of {world}")
    if rank == 0:
        msgs = g.messages_per_sweep() * sweeps
        per_launch_s = kern_ms * 1e-3 / max(launches, 1)
        achieved = g.algorithmic_bytes_per_sweep() / per_launch_s / 1e9
        out = {
            "metric": "edge-messages/sec to BP convergence", "value": msgs / dt, "unit": "edge-messages/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.rows}x{a.cols} 2D-grid BN, k=4, {g.n} nodes, {g.n_edges} edges, cut into "
                                   f"{world} row stripes (BASELINE.json configs[3]), {ev.ne} evidence nodes, eps={a.eps:g}",
                       "sweeps_per_step": sweeps / a.steps, "messages_per_sweep": g.messages_per_sweep(),
                       "parallelism": (f"edge-cut x{world}, halo exchange inside the resident kernel: cut-edge halves stored into the "
                                       f"peer's exchange region over xGMI, per-tile generation granules, per-rank residual granules "
                                       f"(one launch per rank and run, no collective)") if in_kernel else
                                      (f"edge-cut x{world}, 1 in-place RCCL all-gather per sweep ({seg} B per rank incl. residual slots)"),
                       "exchange": "in-kernel (peer-mapped memory)" if in_kernel else "rccl all-gather per sweep",
                       # what the RCCL communicator itself reports (0: none was created -- BN_NO_RCCL)
                       "rccl_ranks": rccl_ranks, "world_size": world, "in_kernel_exchange_verified": bool(verified)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0 * world, "unit": "GB/s",
                         "frac": achieved / (8000.0 * world), "traffic": None,
                         "kernel": "bp_sweep_kernel (interior | cut-touching tiles) + all-gather on a second stream"
                                   if path == 0 else "bp_resident_kernel, dataflow form with peer stores",
                         "avg_launch_us": per_launch_s * 1e6,
                         "avg_launch_us_source": "device clock, sweep start to next sweep start (exchange included)"},
        }
        if rccl:
            out["rccl_exchange"] = rccl
        if other:
            out["in_kernel_exchange"] = other
    eng.close()
    # weak scaling: 316 rows per GPU
    if not getattr(a, "no_weak", False):
        gw = synth.grid(a.rows * world, a.cols, 4, seed=2)
        evw = synth.random_evidence(gw, a.evidence, seed=7)
        engw = make_shard(gw, rank, world, device, in_kernel=not os.environ.get("BN_NO_PEER_EXCHANGE"))
        engw.bp_set_evidence(evw)
        in_kernel_w = verify_in_kernel_exchange(engw, gw, evw, a.eps, device, repeats=1) and peer_headline
        engw.set_option("multisweep", 1 if in_kernel_w else 0)
        if in_kernel_w or have_rccl:
            dtw, sw, kw, lw, okw = _timed_runs(engw, a.eps, max(a.steps // 2, 3), 2, dist, torch)
            if not okw and have_rccl:
                in_kernel_w = False
                engw.set_option("multisweep", 0)
                dtw, sw, kw, lw, okw = _timed_runs(engw, a.eps, max(a.steps // 2, 3), 2, dist, torch)
        elif rank == 0:  # several ranks sharing ONE device (test arrangement): their kernels do not fit the chip together
            out["weak_scaling"] = {"error": "in-kernel exchange not available at this size and no RCCL communicator"}
        if rank == 0 and (in_kernel_w or have_rccl):
            out["weak_scaling"] = {"workload": f"{a.rows * world}x{a.cols} grid, {a.rows} rows per GPU",
                                   "value": gw.messages_per_sweep() * sw / dtw, "unit": "edge-messages/s",
                                   "avg_sweep_plus_exchange_us": kw * 1e3 / max(lw, 1),
                                   "exchange": "in-kernel (peer-mapped memory)" if in_kernel_w else "rccl all-gather per sweep",
                                   "sweeps_per_step": sw / max(a.steps // 2, 3)}
        engw.close()
    # the other way to use N GPUs on a network that fits one: every GPU holds the whole network and answers
    # its own queries (a different evidence set per rank); no exchange, no collective.  Reported beside the
    # edge-cut figure, never as `value` (BASELINE.json configs[3] is the partitioned grid).
    if not getattr(a, "no_replicas", False):
        evr = synth.random_evidence(g, a.evidence, seed=7 + rank)
        with Engine(g, device=device) as er:
            er.bp_set_evidence(evr)
            dtr, sr, _, _, _ = _timed_runs(er, a.eps, a.steps, a.warmup, dist, torch)
            rpath = er.last_path()
        tot = torch.tensor([float(sr)], dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        if rank == 0:
            out["replicated_queries"] = {
                "workload": f"{world} independent queries at a time: the whole {a.rows}x{a.cols} grid on every GPU, "
                            f"a different evidence set per rank, no collective",
                "value": g.messages_per_sweep() * float(tot[0]) / dtr, "unit": "edge-messages/s",
                "ms_per_query": dtr / a.steps * 1e3, "path": "resident tiles, one launch per run" if rpath == 2 else "one launch per sweep"}
    dist.barrier()
    dog.cancel()
    _finish(dist)
    if rank == 0:
        benchline.emit(out)   # the contract line is the LAST thing this job prints


def _finish(dist) -> None:
    """Leave the control plane before the result is printed: whatever the process group has to say on shutdown comes first."""
    try:
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


def bench_lw_main(a, rank: int, world: int, local_rank: int) -> None:
    """bench.py --workload lw --gpus N: BASELINE.json configs[4] -- every GPU draws its share of the
    samples (disjoint sample ids = disjoint streams), ONE RCCL all-reduce sums the histograms.
    Weak scaling: a.samples per GPU and step."""
    import threading
    import torch
    limit = float(os.environ.get("BN_BENCH_WATCHDOG_S", "600"))
    dog = threading.Timer(limit, lambda: (print(f"[bench] rank {rank}: exceeded {limit:.0f} s", flush=True), os._exit(124)))
    dog.daemon = True
    dog.start()
    dist = init_control_plane()
    torch.cuda.set_device(local_rank)
    d = synth.random_dag(10000, 4, 64, 4, seed=1)
    ev = synth.random_evidence(d, a.evidence, seed=7).hard_states(d)
    eng = make_shard(d, rank, world, local_rank, in_kernel=False)
    total = a.samples * world
    for w in range(max(a.warmup, 1)):
        eng.lw_run_allreduce(ev, total, seed=1, sample_begin=w * total)
    torch.cuda.synchronize()
    dist.barrier()
    steps = max(1, min(a.steps, 10))
    t0 = time.perf_counter()
    for i in range(steps):
        eng.lw_run_allreduce(ev, total, seed=1, sample_begin=(i + a.warmup) * total)
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        rate = total * steps / float(dt[0])
        out = {
            "metric": "weighted samples/sec (likelihood weighting)", "value": rate, "unit": "samples/s",
            "n_gpus": world, "steps": steps, "warmup": a.warmup, "ms_per_step": float(dt[0]) / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"likelihood weighting, 10 k-node random DAG, {int((ev >= 0).sum())} evidence nodes, "
                                   f"{a.samples} samples per GPU and step (BASELINE.json configs[4])",
                       "parallelism": f"sample ranges x{world}, one RCCL all-reduce of {int(d.k.sum())} doubles per step",
                       "node_samples_per_s": rate * d.n, "rccl_ranks": int(eng.info("rccl_ranks")), "world_size": world}}
    eng.close()
    dist.barrier()
    dog.cancel()
    _finish(dist)
    if rank == 0:
        benchline.emit(out)

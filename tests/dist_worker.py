"""Worker of tests/test_dist_cpu.py: one rank of a world_size-N gloo job on CPU.

Exercises the N>1 control flow and the EXCHANGE LAYOUT of the product's partition plan without a
GPU: each rank computes only what it owns (oracle/bp_oracle.c partial sweep -- checker code, used
here in tests only), writes the message halves it produces into its exchange segment at the
offsets the plan (libbn_mi355x.so, host-only engine) assigns, the segments are all-gathered with
torch.distributed(gloo) exactly like the device path does with RCCL, and every rank reads the
halves it needs from the gathered region.  The result must equal the unsharded oracle bit for bit."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import oracle  # noqa: E402
from bayesiannetwork_amd import _lib, synth  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402


def main():
    case = sys.argv[1]
    overlapped = len(sys.argv) > 2 and sys.argv[2] == "overlapped"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    if case == "grid":
        model, owner_arg = synth.grid(20, 17, 4, seed=3), None
    else:
        model = synth.random_dag(400, 4, 32, [2, 3, 4], seed=8)
        owner_arg = (synth.splitmix64(5, 0, model.n) % np.uint64(world)).astype(np.int32)
    ev = synth.random_evidence(model, 0.05, seed=2)
    eps = 1e-6

    # ---- the product's partition plan (host only: no GPU here, and it does no compute)
    eng = Engine(model, device=_lib.BN_DEVICE_HOST_ONLY, rank=rank, nranks=world, owner=owner_arg)
    li = eng.layout()
    gbase, seg_d2 = li["exchange_base"], li["segment_bytes"] // 16
    ref_pi, ref_lam = eng.edge_refs()
    owned = eng.node_slots() >= 0
    t = torch.from_numpy(owned.astype(np.int32) * (rank + 1))
    dist.all_reduce(t)                      # every node owned exactly once -> owner map on every rank
    owner = (t.numpy() - 1).astype(np.int32)
    assert (owner >= 0).all() and (owner < world).all()
    digest = torch.tensor([li["segment_bytes"], int(owner.sum())], dtype=torch.int64)
    alld = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(alld, digest)
    assert all(torch.equal(d, alld[0]) for d in alld), "ranks disagree on the exchange layout"

    # ---- oracle state + owned-part sweeps
    L = oracle.lib()
    L.oracle_bp_open.restype = ctypes.c_void_p
    L.oracle_bp_array.restype = ctypes.POINTER(ctypes.c_double)
    L.oracle_bp_sweep_owned.restype = ctypes.c_double
    p = lambda a, ct: a.ctypes.data_as(ctypes.POINTER(ct))  # noqa: E731
    h = ctypes.c_void_p(L.oracle_bp_open(model.n, p(model.k, ctypes.c_int32), p(model.in_ptr, ctypes.c_int32),
                                         p(model.in_idx, ctypes.c_int32), p(model.cpt_off, ctypes.c_int64),
                                         p(model.cpt, ctypes.c_double)))
    L.oracle_bp_reset(h, ev.ne, p(ev.node, ctypes.c_int32), p(ev.off, ctypes.c_int32), p(ev.val, ctypes.c_double))
    nm = int(model.msg_off[-1])
    arr = lambda which, n: np.ctypeslib.as_array(L.oracle_bp_array(h, which), shape=(n,))  # noqa: E731
    child = np.repeat(np.arange(model.n), np.diff(model.in_ptr))
    cut = np.nonzero(owner[model.in_idx] != owner[child])[0]
    moff, kpar = model.msg_off, model.k[model.in_idx]

    def pack(npim, nlkm, md):
        """halves this rank produced, at the plan's offsets inside ITS segment; the residual rides in the slots"""
        seg = np.zeros(seg_d2 * 2, dtype=np.float64)
        for e in cut:
            a, b = owner[model.in_idx[e]], owner[child[e]]
            if a == rank:
                o = (int(ref_pi[e]) - gbase - rank * seg_d2) * 2
                seg[o:o + kpar[e]] = npim[moff[e]:moff[e + 1]]
            if b == rank:
                o = (int(~ref_lam[e]) - gbase - rank * seg_d2) * 2
                seg[o:o + kpar[e]] = nlkm[moff[e]:moff[e + 1]]
        seg[-256:] = 0.0
        seg[-256] = md
        return seg

    def unpack(G, pim, lkm):
        """halves produced by the other endpoint's owner, from the gathered region"""
        for e in cut:
            a, b = owner[model.in_idx[e]], owner[child[e]]
            if b == rank:
                o = (int(ref_pi[e]) - gbase) * 2
                pim[moff[e]:moff[e + 1]] = G[o:o + kpar[e]]
            if a == rank:
                o = (int(~ref_lam[e]) - gbase) * 2
                lkm[moff[e]:moff[e + 1]] = G[o:o + kpar[e]]
        return max(float(G[(q + 1) * seg_d2 * 2 - 256]) for q in range(world))

    sweeps = 0
    if len(sys.argv) > 2 and sys.argv[2] == "granules":
        # The IN-KERNEL exchange of the sharded resident kernel (bn_resident.hip, SHARD) with gloo standing in for the
        # peer-mapped memory: nobody gathers segments.  A rank PUSHES each half it produced for a cut edge to the one rank
        # that reads it (the owner across the cut: the segment the edge's other half lives in), pushes the generation
        # granule of every tile that touches a cut to the ranks in the tile's report mask, and its residual to every
        # rank; before a sweep it checks -- like the kernel's poll -- that every neighbour slot of every one of its
        # tiles (the table bn_peer_import built from the ranks' blobs) carries the previous sweep's generation, and it
        # reads nothing but its own copy of the exchange region.
        blobs = [None] * world
        dist.all_gather_object(blobs, eng.peer_export())
        eng.peer_import(blobs)                      # host-only engine: tables only
        nbr, pub = eng.flow_tables()
        tiles = eng.node_tiles()
        region = np.zeros(world * seg_d2 * 2, dtype=np.float64)      # this rank's exchange region
        table = {}                                                    # slot -> generation (this rank's granule table)
        SLOTS = 2048
        while True:
            md = L.oracle_bp_sweep_owned(h, p(owner, ctypes.c_int32), rank)
            npim, nlkm = arr(6, nm), arr(7, nm)
            gen = sweeps + 1
            outbox = [{"at": [], "val": [], "gran": [], "res": md} for _ in range(world)]
            for e in cut:
                u, v = model.in_idx[e], child[e]
                a, b = owner[u], owner[v]
                if a == rank:   # pi half: local copy + the child's owner
                    o = (int(ref_pi[e]) - gbase) * 2
                    region[o:o + kpar[e]] = npim[moff[e]:moff[e + 1]]
                    outbox[b]["at"].append(o); outbox[b]["val"].append(npim[moff[e]:moff[e + 1]].copy())
                if b == rank:   # lambda half: local copy + the parent's owner
                    o = (int(~ref_lam[e]) - gbase) * 2
                    region[o:o + kpar[e]] = nlkm[moff[e]:moff[e + 1]]
                    outbox[a]["at"].append(o); outbox[a]["val"].append(nlkm[moff[e]:moff[e + 1]].copy())
            for t in range(nbr.shape[0]):
                table[rank * SLOTS + t] = gen
                for q in range(world):
                    if (int(pub[t]) >> q) & 1:
                        outbox[q]["gran"].append(rank * SLOTS + t)
            inbox = [None] * world
            dist.all_gather_object(inbox, outbox)    # transport only: rank r takes inbox[q][r], what q addressed to it
            res = 0.0
            for q in range(world):
                box = inbox[q][rank]
                res = max(res, box["res"])           # every rank's residual reaches every rank
                if q == rank:
                    continue
                for o, val in zip(box["at"], box["val"]):
                    region[o:o + val.size] = val
                for slot in box["gran"]:
                    table[slot] = gen
            # the poll of the next sweep: every neighbour tile, local or across a cut, has finished this one
            for t in range(nbr.shape[0]):
                for slot in nbr[t][nbr[t] >= 0]:
                    assert table.get(int(slot)) == gen, f"rank {rank} tile {t}: neighbour slot {slot} never reported sweep {gen}"
            unpack(region, npim, nlkm)               # reads this rank's own region only
            L.oracle_bp_commit(h)
            sweeps += 1
            if res < eps:
                break
        # a tile reports to a rank exactly when it has a neighbour there, and the lists mirror each other across the cut
        alltabs = [None] * world
        dist.all_gather_object(alltabs, (nbr, pub))
        for t in range(nbr.shape[0]):
            want_mask = 0
            for slot in nbr[t][nbr[t] >= 0]:
                q, t2 = divmod(int(slot), SLOTS)
                if q != rank:
                    want_mask |= 1 << q
                    assert rank * SLOTS + t in alltabs[q][0][t2] and (int(alltabs[q][1][t2]) >> rank) & 1
            assert int(pub[t]) == want_mask
        assert np.array_equal(np.unique(tiles[(owner == rank)]), np.arange(nbr.shape[0]))
    elif not overlapped:
        while True:
            md = L.oracle_bp_sweep_owned(h, p(owner, ctypes.c_int32), rank)
            npim, nlkm = arr(6, nm), arr(7, nm)
            seg = pack(npim, nlkm, md)
            gathered = [torch.zeros(seg.size, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(gathered, torch.from_numpy(seg))  # the device path: in-place ncclAllGather
            res = unpack(torch.cat(gathered).numpy(), npim, nlkm)
            L.oracle_bp_commit(h)
            sweeps += 1
            if res < eps:
                break
    else:
        # The engine's overlapped order (bn_engine.cpp step_sweep_overlapped): the nodes of the plan's
        # INTERIOR tiles of sweep s+1 are computed while the all-gather of sweep s is still in flight
        # (async gloo op), the nodes of the tiles that touch a cut edge after it has landed.  The split
        # is the product plan's (node -> tile < n_interior_tiles); a virtual owner id selects each part.
        tiles = eng.node_tiles()
        part = owner.copy()
        part[(owner == rank) & (tiles >= li["n_interior_tiles"])] = rank + world
        md = max(L.oracle_bp_sweep_owned(h, p(part, ctypes.c_int32), rank),
                 L.oracle_bp_sweep_owned(h, p(part, ctypes.c_int32), rank + world))
        while True:
            seg = pack(arr(6, nm), arr(7, nm), md)
            gathered = [torch.zeros(seg.size, dtype=torch.float64) for _ in range(world)]
            work = dist.all_gather(gathered, torch.from_numpy(seg), async_op=True)
            L.oracle_bp_commit(h)       # the sweep's own results become current; its halo has NOT landed yet
            sweeps += 1
            md_int = L.oracle_bp_sweep_owned(h, p(part, ctypes.c_int32), rank)          # interior tiles of the next sweep
            work.wait()
            res = unpack(torch.cat(gathered).numpy(), arr(2, nm), arr(3, nm))           # halo lands in the CURRENT state
            if res < eps:
                break                   # what the interior launch computed past convergence is simply dropped
            md = max(md_int, L.oracle_bp_sweep_owned(h, p(part, ctypes.c_int32), rank + world))
    bel = np.zeros(int(model.k.sum()))
    off = model.node_off
    buf = (ctypes.c_double * 256)()
    for v in np.nonzero(owner == rank)[0]:
        L.oracle_bp_belief(h, int(v), buf)
        bel[off[v]:off[v + 1]] = np.frombuffer(buf, dtype=np.float64, count=int(model.k[v]))
    tb = torch.from_numpy(bel)
    dist.all_reduce(tb)                                  # what multigpu.gather_beliefs does
    if rank == 0:
        want = oracle.bp_run(model, ev, eps)
        assert sweeps == want["sweeps"], (sweeps, want["sweeps"])
        assert np.array_equal(tb.numpy(), want["beliefs"]), "sharded result differs from the unsharded oracle"
        order = sys.argv[2] if len(sys.argv) > 2 else "plain"
        print(f"DIST_OK case={case} world={world} order={order} sweeps={sweeps} cut_edges={cut.size}")
    L.oracle_bp_close(h)
    dist.barrier()


if __name__ == "__main__":
    main()

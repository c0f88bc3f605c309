// bn_engine_shard.cpp -- sharded engines: the RCCL communicator (loaded lazily), the in-kernel exchange's peer tables, and the
// one-device stand-in for the all-gather that the single-GPU tests use.
#include "bn_engine_internal.hpp"

// Emulates the per-sweep all-gather between `n` shard engines that live on ONE device (tests on
// a single-GPU box): every engine's own segment of the buffer written by `sweep` is copied into
// all the others.  The data path proper uses RCCL (step_exchange).
extern "C" int bn_debug_allgather(bn_engine** engs, int32_t n, int32_t sweep) {
    if (!engs || n < 1) return fail(BN_ERR_ARG, "bad argument");
    for (int32_t i = 0; i < n; ++i)
        if (!engs[i] || engs[i]->host_only || engs[i]->plan.nranks != n || engs[i]->plan.rank != i)
            return fail(BN_ERR_ARG, "engine i must be shard i of n on a device");
    const int buf = (sweep + 1) & 1;
    for (int32_t i = 0; i < n; ++i) HIPCHK(hipStreamSynchronize(engs[i]->stream));
    const Plan& p0 = engs[0]->plan;
    const size_t seg_bytes = size_t(p0.seg_d2) * 16;
    for (int32_t src = 0; src < n; ++src)
        for (int32_t dst = 0; dst < n; ++dst) {
            if (src == dst) continue;
            const Plan& ps = engs[src]->plan;
            const Plan& pd = engs[dst]->plan;
            if (ps.seg_d2 != pd.seg_d2) return fail(BN_ERR_ARG, "shards disagree on the segment size");
            const char* from = reinterpret_cast<const char*>(engs[src]->d_rec[buf] + 2 * ps.g_base) + size_t(src) * seg_bytes;
            char* to = reinterpret_cast<char*>(engs[dst]->d_rec[buf] + 2 * pd.g_base) + size_t(src) * seg_bytes;
            HIPCHK(hipMemcpy(to, from, seg_bytes, hipMemcpyDeviceToDevice));
        }
    HIPCHK(hipDeviceSynchronize());
    return BN_OK;
}

// ---- RCCL communicator ---------------------------------------------------------------------------
extern "C" int bn_comm_unique_id(void* id_out128) {
    if (!id_out128) return fail(BN_ERR_ARG, "null argument");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(BN_ERR_COMM, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r));
    std::memcpy(id_out128, &id, sizeof id);
    return BN_OK;
}

extern "C" int bn_comm_init(bn_engine* e, const void* id128) {
    if (!e || !id128) return fail(BN_ERR_ARG, "null argument");
    if (e->host_only) return fail(BN_ERR_STATE, "host-only engine");
    int rc = load_rccl();
    if (rc) return rc;
    ON_DEVICE(e);
    if (e->comm) { (void)g_rccl.CommDestroy(e->comm); e->comm = nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclResult_t r = g_rccl.CommInitRank(&e->comm, e->plan.nranks, id, e->plan.rank);
    if (r != ncclSuccess) {
        e->comm = nullptr;
        return fail(BN_ERR_COMM, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
    }
    return BN_OK;
}

// ---- in-kernel halo exchange of sharded engines: what a rank tells the others, and what it does with what they say ----
namespace {
struct PeerBlobHeader {
    uint32_t magic;           // 'BNPB'
    int32_t rank, nranks;
    int32_t device;           // HIP ordinal in the exporting process, -1: host-only engine
    int64_t pid;
    int32_t n_boundary;       // (node, tile) pairs that follow
    int32_t shapes_ok;        // this shard's tiles can run in the resident kernel
    int64_t g_base, rec_bytes;  // start of the exchange region in its record buffers (double2 units), size of one buffer
    uint64_t flow_ptr, rec0_ptr, rec1_ptr;  // raw device pointers (valid inside the exporting process)
    hipIpcMemHandle_t h_flow, h_rec0, h_rec1;  // ... and their handles for other processes
};
constexpr uint32_t kPeerBlobMagic = 0x42504e42u;
}  // namespace

extern "C" int64_t bn_peer_blob_size(bn_engine* e) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    return int64_t(sizeof(PeerBlobHeader)) + int64_t(e->plan.boundary_node.size()) * 8;
}

extern "C" int bn_peer_export(bn_engine* e, void* blob, int64_t cap) {
    if (!e || !blob) return fail(BN_ERR_ARG, "null argument");
    if (e->plan.nranks < 2) return fail(BN_ERR_STATE, "not a sharded engine");
    if (cap < bn_peer_blob_size(e)) return fail(BN_ERR_ARG, "blob buffer too small (bn_peer_blob_size)");
    PeerBlobHeader h;
    std::memset(&h, 0, sizeof h);
    h.magic = kPeerBlobMagic;
    h.rank = e->plan.rank;
    h.nranks = e->plan.nranks;
    h.device = e->host_only ? -1 : e->device;
    h.pid = int64_t(getpid());
    h.n_boundary = int32_t(e->plan.boundary_node.size());
    h.shapes_ok = (e->host_only || e->shard_shapes_ok) ? 1 : 0;
    h.g_base = e->plan.g_base;
    h.rec_bytes = e->plan.rec_total_doubles * 8;
    if (!e->host_only && e->shard_shapes_ok) {
        ON_DEVICE(e);
        h.flow_ptr = uint64_t(reinterpret_cast<uintptr_t>(e->d_flow));
        h.rec0_ptr = uint64_t(reinterpret_cast<uintptr_t>(e->d_rec[0]));
        h.rec1_ptr = uint64_t(reinterpret_cast<uintptr_t>(e->d_rec[1]));
        HIPCHK(hipIpcGetMemHandle(&h.h_flow, e->d_flow));
        HIPCHK(hipIpcGetMemHandle(&h.h_rec0, e->d_rec[0]));
        HIPCHK(hipIpcGetMemHandle(&h.h_rec1, e->d_rec[1]));
    }
    char* out = static_cast<char*>(blob);
    std::memcpy(out, &h, sizeof h);
    int32_t* pairs = reinterpret_cast<int32_t*>(out + sizeof h);
    for (int32_t i = 0; i < h.n_boundary; ++i) {
        pairs[2 * i] = e->plan.boundary_node[i];
        pairs[2 * i + 1] = e->plan.boundary_tile[i];
    }
    return BN_OK;
}

// blobs[r] = what rank r exported (this rank's own entry included), r = 0 .. nranks - 1
extern "C" int bn_peer_import(bn_engine* e, const void* const* blobs, const int64_t* sizes, int32_t n) {
    if (!e || !blobs || !sizes) return fail(BN_ERR_ARG, "null argument");
    Plan& p = e->plan;
    if (p.nranks < 2) return fail(BN_ERR_STATE, "not a sharded engine");
    if (n != p.nranks) return fail(BN_ERR_ARG, "one blob per rank");
    std::vector<PeerBlobHeader> hd(n);
    std::vector<std::unordered_map<int32_t, int32_t>> tile_of(n);  // per rank: boundary node -> tile
    bool all_ok = true;
    for (int32_t r = 0; r < n; ++r) {
        if (!blobs[r] || sizes[r] < int64_t(sizeof(PeerBlobHeader))) return fail(BN_ERR_ARG, "short peer blob");
        std::memcpy(&hd[r], blobs[r], sizeof(PeerBlobHeader));
        if (hd[r].magic != kPeerBlobMagic || hd[r].rank != r || hd[r].nranks != n) return fail(BN_ERR_ARG, "peer blob of the wrong rank / world");
        if (sizes[r] < int64_t(sizeof(PeerBlobHeader)) + int64_t(hd[r].n_boundary) * 8) return fail(BN_ERR_ARG, "short peer blob");
        const int32_t* pairs = reinterpret_cast<const int32_t*>(static_cast<const char*>(blobs[r]) + sizeof(PeerBlobHeader));
        for (int32_t i = 0; i < hd[r].n_boundary; ++i) tile_of[r][pairs[2 * i]] = pairs[2 * i + 1];
        all_ok = all_ok && hd[r].shapes_ok != 0;
    }
    // neighbour tiles across the cut and the ranks each tile reports to
    const int32_t nt = int32_t(p.tiles.size());
    std::vector<std::vector<int32_t>> remote(nt);
    e->pub_mask.assign(std::max(nt, 1), 0u);
    for (const Plan::CutLink& c : p.cut_links) {
        auto it = tile_of[c.rank].find(c.node);
        if (it == tile_of[c.rank].end()) return fail(BN_ERR_ARG, "a peer blob does not list the node across a cut edge (different model or partition?)");
        if (it->second < 0 || it->second >= kFlowSlotsPerRank) { all_ok = false; continue; }
        remote[c.tile].push_back(c.rank * kFlowSlotsPerRank + it->second);
        e->pub_mask[c.tile] |= 1u << c.rank;
    }
    for (auto& v : remote) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
    }
    const std::string err = build_neighbour_table(p, remote);
    if (!err.empty()) all_ok = false;
    e->shard_flow_ok = false;
    if (e->host_only) return BN_OK;  // tables only (tests)
    if (!all_ok || !e->shard_shapes_ok) return BN_OK;  // stays on the per-sweep launches + RCCL
    ON_DEVICE(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    std::vector<PeerTable> peers(n);
    for (int32_t r = 0; r < n; ++r) {
        if (r == p.rank) {
            peers[r] = PeerTable{e->d_flow, e->d_rec[0], e->d_rec[1], p.g_base, p.rec_total_doubles * 8};
        } else if (hd[r].pid == int64_t(getpid())) {  // another engine of this process: its pointers are ours
            if (hd[r].device != e->device) {
                hipError_t pe = hipDeviceEnablePeerAccess(hd[r].device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                    return fail(BN_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe));
                (void)hipGetLastError();
            }
            peers[r] = PeerTable{reinterpret_cast<FlowSync*>(uintptr_t(hd[r].flow_ptr)), reinterpret_cast<double*>(uintptr_t(hd[r].rec0_ptr)),
                                 reinterpret_cast<double*>(uintptr_t(hd[r].rec1_ptr)), hd[r].g_base, hd[r].rec_bytes};
        } else {
            void* q[3] = {nullptr, nullptr, nullptr};
            const hipIpcMemHandle_t hs[3] = {hd[r].h_flow, hd[r].h_rec0, hd[r].h_rec1};
            for (int k = 0; k < 3; ++k) {
                HIPCHK(hipIpcOpenMemHandle(&q[k], hs[k], hipIpcMemLazyEnablePeerAccess));
                e->ipc_opened.push_back(q[k]);
            }
            peers[r] = PeerTable{static_cast<FlowSync*>(q[0]), static_cast<double*>(q[1]), static_cast<double*>(q[2]), hd[r].g_base, hd[r].rec_bytes};
        }
    }
    if (e->d_peers) { (void)hipFree(e->d_peers); e->d_peers = nullptr; }
    if (e->d_pub_mask) { (void)hipFree(e->d_pub_mask); e->d_pub_mask = nullptr; }
    if (e->d_nbr) { (void)hipFree(e->d_nbr); e->d_nbr = nullptr; }
    int rc;
    if ((rc = upload(&e->d_peers, peers, e->stream))) return rc;
    if ((rc = upload(&e->d_pub_mask, e->pub_mask, e->stream))) return rc;
    if ((rc = upload(&e->d_nbr, p.nbr, e->stream))) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    e->shard_flow_ok = true;
    return BN_OK;
}

// host copies of the dataflow tables (tests): nbr_out [n_tiles * nbr_chunks * 64] (bn_get_info "nbr_chunks"), pub_out [n_tiles]
extern "C" int bn_layout_flow(bn_engine* e, int32_t* nbr_out, uint32_t* pub_out) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (nbr_out) std::copy(e->plan.nbr.begin(), e->plan.nbr.end(), nbr_out);
    if (pub_out) {
        for (size_t t = 0; t < e->plan.tiles.size(); ++t) pub_out[t] = t < e->pub_mask.size() ? e->pub_mask[t] : 0u;
    }
    return BN_OK;
}


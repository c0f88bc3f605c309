// bn_engine.cpp -- C ABI (include/bn_mi355x.h) over the HIP kernels: device memory, evidence, the run loop of belief propagation
// for a single query (its steps, the one-launch execution paths and their dispatch), options, read-out.  Batches, sharding, introspection
// and the samplers' entry points: bn_engine_batch.cpp, bn_engine_shard.cpp, bn_engine_tools.cpp (bn_engine_internal.hpp has the map).
#include "bn_engine_internal.hpp"

#include <mutex>

thread_local std::string bn_eng::g_err;
RcclApi bn_eng::g_rccl;

int bn_eng::fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

int bn_eng::load_rccl() {
    if (g_rccl.handle) return BN_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(BN_ERR_COMM, std::string("cannot load librccl: ") + dlerror());
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(h, "ncclCommCount"));
    if (!a.AllReduce || !a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString)
        return fail(BN_ERR_COMM, "librccl lacks a required symbol");
    g_rccl = a;
    return BN_OK;
}


// Streams of destroyed engines, kept for the next bn_create on the same device: creating a non-blocking stream costs 1.5-4 ms on an
// MI355X box (scripts/experiments/create_split.py) -- most of what constructing a functor on an ALARM-sized network takes once the
// runtime is up.  A parked stream is idle (free_engine synchronises it first); at most eight per process.
namespace {
struct ParkedStream { int device; hipStream_t stream; };
std::mutex g_stream_mu;
std::vector<ParkedStream> g_parked_streams;
hipStream_t take_parked_stream(int device) {
    std::lock_guard<std::mutex> lock(g_stream_mu);
    for (size_t i = 0; i < g_parked_streams.size(); ++i)
        if (g_parked_streams[i].device == device) {
            hipStream_t s = g_parked_streams[i].stream;
            g_parked_streams.erase(g_parked_streams.begin() + i);
            return s;
        }
    return nullptr;
}
// The FIRST device engine of a process on a device also parks a few spare streams: with another engine's stream alive, creating one more
// costs 5-11 ms (scripts/experiments/create_split_alive.py) -- nearly all of a functor's construction on an ALARM-sized network --
// while that first bn_create spends 80-240 ms bringing the runtime up anyway.
constexpr int kSpareStreams = 3;
std::vector<int> g_primed_devices;
void prime_spare_streams(int device) {
    {
        std::lock_guard<std::mutex> lock(g_stream_mu);
        for (int d : g_primed_devices)
            if (d == device) return;
        g_primed_devices.push_back(device);
    }
    if (std::getenv("BN_NO_SPARE_STREAMS")) return;
    for (int i = 0; i < kSpareStreams; ++i) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return; }
        std::lock_guard<std::mutex> lock(g_stream_mu);
        g_parked_streams.push_back(ParkedStream{device, s});
    }
}
void park_stream(int device, hipStream_t s) {
    if (hipStreamSynchronize(s) == hipSuccess) {
        std::lock_guard<std::mutex> lock(g_stream_mu);
        if (g_parked_streams.size() < 8) { g_parked_streams.push_back(ParkedStream{device, s}); return; }
    }
    (void)hipStreamDestroy(s);
}
}  // namespace

void bn_eng::free_engine(bn_engine* e) {
    if (!e) return;
    if (e->dense) { free_engine(e->dense); e->dense = nullptr; }
    if (!e->host_only) {
        DeviceGuard guard;
        (void)guard.enter(e->device);
        lw_free(e->lw);
        for (void* q : e->ipc_opened) (void)hipIpcCloseMemHandle(q);
        if (e->d_peers) (void)hipFree(e->d_peers);
        if (e->d_pub_mask) (void)hipFree(e->d_pub_mask);
        if (e->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(e->comm);
        if (e->ev_swept) (void)hipEventDestroy(e->ev_swept);
        if (e->ev_gathered) (void)hipEventDestroy(e->ev_gathered);
        if (e->comm_stream) (void)hipStreamDestroy(e->comm_stream);
        void* ptrs[] = {e->d_tiles, e->d_classes, e->d_flat_tab, e->d_cpt, e->d_rec[0], e->d_rec[1], e->d_node[0], e->d_node[1],
                        e->d_out, e->d_frozen, e->d_slot_node, e->d_slot_boff, e->d_node_tile, e->d_node_nl,
                        e->d_inrefs, e->d_res_hist, e->d_ctl, e->d_beliefs, e->d_ev, e->d_rsync, e->d_flow, e->d_nbr,
                        e->d_s_ent, e->d_s_cpt, e->d_s_term, e->d_s_clist, e->d_s_bslot, e->d_s_cslot, e->d_s_nvidx, e->d_s_nvslot, e->d_s_init, e->d_s_state, e->d_s_nodeoff,
                        e->d_m_parts, e->d_m_ent, e->d_m_cpt, e->d_m_term, e->d_m_clist, e->d_m_bslot, e->d_m_cslot, e->d_m_nvidx, e->d_m_nvslot,
                        e->d_m_init, e->d_m_nodeoff, e->d_m_msgfirst, e->d_m_state, e->d_m_frz, e->d_m_sync,
                        e->d_g_tiles, e->d_g_slotptr, e->d_g_cnode, e->d_g_pitem, e->d_g_oedge, e->d_g_eperm, e->d_g_nperm, e->d_g_cpt, e->d_g_init, e->d_g_state, e->d_g_frz, e->d_g_sync, e->d_g_k, e->d_g_inptr, e->d_g_inidx, e->d_g_noff, e->d_g_nbr, e->d_g_flow,
                        e->batch.d_rec[0], e->batch.d_rec[1], e->batch.d_node[0], e->batch.d_node[1], e->batch.d_frozen,
                        e->batch.d_beliefs, e->batch.d_res_hist, e->batch.d_sync, e->batch.d_ev, e->batch.d_ctl, e->batch.d_s_state,
                        e->batch.d_g_state, e->batch.d_g_frz, e->batch.d_g_sync};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        if (e->h_ctl) (void)hipHostFree(e->h_ctl);
        if (e->h_abort) (void)hipHostFree(e->h_abort);
        if (e->batch.h_ctl) (void)hipHostFree(e->batch.h_ctl);
        if (e->batch.h_ev) (void)hipHostFree(e->batch.h_ev);
        if (e->batch.h_beliefs) (void)hipHostFree(e->batch.h_beliefs);
        if (e->h_ev) (void)hipHostFree(e->h_ev);
        if (e->h_beliefs) (void)hipHostFree(e->h_beliefs);
        for (hipEvent_t ev : e->events) (void)hipEventDestroy(ev);
        if (e->stream) park_stream(e->device, e->stream);
    }
    delete e;
}

extern "C" const char* bn_last_error(void) { return g_err.c_str(); }
extern "C" const char* bn_version(void) { return "bn_mi355x 0.1 (gfx950)"; }

static void debug_segv_handler(int sig) {
    void* frames[64];
    int n = backtrace(frames, 64);
    const char msg[] = "[bn_mi355x] fatal signal, native backtrace:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

// state slots of the mid-size kernel (bn_mid.hip): per slot the four double-buffered state arrays, the marks, the barrier words
int bn_eng::mid_reserve_slots(bn_engine* e, int32_t slots) {
    if (slots <= e->mid_slots) return BN_OK;
    const SmallPlan& g0 = e->mid.parts[0];
    if (e->stream) HIPCHK(hipStreamSynchronize(e->stream));
    if (e->d_m_state) (void)hipFree(e->d_m_state);
    if (e->d_m_frz) (void)hipFree(e->d_m_frz);
    if (e->d_m_sync) (void)hipFree(e->d_m_sync);
    e->d_m_state = nullptr; e->d_m_frz = nullptr; e->d_m_sync = nullptr; e->mid_slots = 0;
    int r;
    if ((r = dalloc(&e->d_m_state, size_t(slots) * size_t(4 * g0.M + 4 * g0.N)))) return r;
    if ((r = dalloc(&e->d_m_frz, size_t(slots) * size_t(g0.N)))) return r;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->d_m_sync), size_t(slots) * kMidSyncBytes));
    e->mid_slots = slots;
    return BN_OK;
}

// The register-resident DAG path's full plan (padded CPT image), its device tables and their upload -- at the first use of the path.
// bn_create builds the LIGHT plan only (tile tables + the features dag_applies reads): on the 316 x 316 grid, where the resident tiles
// are the default, the image alone is 51 MB and 140 ms of host work nobody asked for (missing #5).
int bn_eng::ensure_dag(bn_engine* e) {
    if (!e->dag.ok) return fail(BN_ERR_STATE, "not eligible for the register-resident DAG path: " + (e->dag.why.empty() ? std::string("disabled") : e->dag.why));
    if (e->dag.light) {
        try {
            DagPlan full;
            build_dag_plan(e->plan, e->dag_cap, full, false);
            if (!full.ok || full.tiles.size() != e->dag.tiles.size() || full.blocks != e->dag.blocks || full.slot_ptr != e->dag.slot_ptr)
                return fail(BN_ERR_STATE, "the register-resident DAG plan changed between its light and its full build");
            e->dag = std::move(full);
            build_dag_device_tables(e->dag, e->dag_tables);
        } catch (const std::bad_alloc&) {
            return fail(BN_ERR_ALLOC, "out of host memory while building the register-resident DAG plan");
        }
    }
    if (e->host_only || e->dag_ready) return BN_OK;
    ON_DEVICE(e);
    const Plan& p = e->plan;
    const DagPlan& dp = e->dag;
    int r2;
    {   // (an earlier attempt that failed half-way -- out of device memory -- must not leak what it had got)
        void** mine[] = {reinterpret_cast<void**>(&e->d_g_slotptr), reinterpret_cast<void**>(&e->d_g_tiles), reinterpret_cast<void**>(&e->d_g_cnode),
                         reinterpret_cast<void**>(&e->d_g_pitem), reinterpret_cast<void**>(&e->d_g_oedge), reinterpret_cast<void**>(&e->d_g_eperm),
                         reinterpret_cast<void**>(&e->d_g_nperm), reinterpret_cast<void**>(&e->d_g_cpt), reinterpret_cast<void**>(&e->d_g_init),
                         reinterpret_cast<void**>(&e->d_g_k), reinterpret_cast<void**>(&e->d_g_inptr), reinterpret_cast<void**>(&e->d_g_inidx),
                         reinterpret_cast<void**>(&e->d_g_noff), reinterpret_cast<void**>(&e->d_g_state), reinterpret_cast<void**>(&e->d_g_frz),
                         reinterpret_cast<void**>(&e->d_g_sync), reinterpret_cast<void**>(&e->d_g_nbr), reinterpret_cast<void**>(&e->d_g_flow)};
        for (void** q : mine)
            if (*q) { (void)hipFree(*q); *q = nullptr; }
    }
    if ((r2 = upload(&e->d_g_slotptr, dp.slot_ptr, e->stream))) return r2;
    {
        const DagDeviceTables& dt = e->dag_tables;
        if ((r2 = upload(&e->d_g_tiles, dt.tiles, e->stream))) return r2;
        if ((r2 = upload(&e->d_g_cnode, dp.cnode, e->stream))) return r2;
        if ((r2 = upload(&e->d_g_pitem, dt.pitem, e->stream))) return r2;
        if ((r2 = upload(&e->d_g_oedge, dt.oedge, e->stream))) return r2;
        if ((r2 = upload(&e->d_g_eperm, dt.eperm, e->stream))) return r2;
        if ((r2 = upload(&e->d_g_nperm, dt.nperm, e->stream))) return r2;
    }
    if ((r2 = upload(&e->d_g_cpt, dp.cpt_img, e->stream))) return r2;
    if ((r2 = upload(&e->d_g_init, dp.npi_init, e->stream))) return r2;
    if (!dp.uniform4) {
        if ((r2 = upload(&e->d_g_k, p.k, e->stream))) return r2;
        if ((r2 = upload(&e->d_g_inptr, p.in_ptr, e->stream))) return r2;
        if ((r2 = upload(&e->d_g_inidx, p.in_idx, e->stream))) return r2;
        if ((r2 = upload(&e->d_g_noff, p.node_off, e->stream))) return r2;
    }
    const size_t sd = size_t(dag_state_doubles(dp.E, dp.n));
    if ((r2 = dalloc(&e->d_g_state, sd))) return r2;
    HIPCHK(hipMemsetAsync(e->d_g_state, 0, std::max<size_t>(sd, 1) * 8, e->stream));
    if ((r2 = dalloc(&e->d_g_frz, size_t(dp.n)))) return r2;
    HIPCHK(hipMemsetAsync(e->d_g_frz, 0, size_t(dp.n), e->stream));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->d_g_sync), sizeof(ResidentSync)));
    {   // the dataflow form of a single query (bn_dag.hip): neighbour tiles, granule table + verdict words
        DagFlowTables ft;
        try { build_dag_flow_tables(dp, p, ft); } catch (const std::bad_alloc&) { ft = DagFlowTables(); }
        e->dag_flow_ok = ft.ok && int64_t(dp.blocks) + 1 <= int64_t(e->n_cus);   // (+ the service block: every block co-resident)
        e->dag_flow_max_nbr = ft.max_nbr;
        if (e->dag_flow_ok) {
            if ((r2 = upload(&e->d_g_nbr, ft.nbr, e->stream))) return r2;
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->d_g_flow), dag_flow_sync_bytes(dp.tiles.size())));
            HIPCHK(hipStreamSynchronize(e->stream));   // (`ft` is a local)
        }
    }
    HIPCHK(hipStreamSynchronize(e->stream));   // (upload() copies from the plan's vectors: they stay, but the order against a reload is then plain)
    e->dag_sync_dirty = true;
    e->dag_ev_applied = false;
    e->dag_ready = true;
    return BN_OK;
}

static int create_impl(const bn_model_desc* desc, const ShardSpec& shard, bn_engine** out) {
    if (!desc || !out) return fail(BN_ERR_ARG, "null argument");
    if (std::getenv("BN_DEBUG")) signal(SIGSEGV, debug_segv_handler);
    *out = nullptr;
    bn_engine* e = new (std::nothrow) bn_engine();
    if (!e) return fail(BN_ERR_ALLOC, "out of host memory");
    if (const char* t = std::getenv("BN_TIMING")) e->timing = std::atoi(t) != 0;  // default off, see bn_engine::timing
    std::string err;
    auto t_phase = std::chrono::steady_clock::now();
    auto lap = [&](int which) {   // bn_get_info "create_us_*": where the construction of this engine went
        const auto now = std::chrono::steady_clock::now();
        e->create_us[which] += std::chrono::duration_cast<std::chrono::microseconds>(now - t_phase).count();
        t_phase = now;
    };
    try {
        err = build_plan(*desc, shard, e->plan);
    } catch (const std::bad_alloc&) {
        delete e;
        return fail(BN_ERR_ALLOC, "out of host memory while building the layout plan");
    }
    if (!err.empty()) {
        delete e;
        return fail(BN_ERR_ARG, err);
    }
    const Plan& p = e->plan;
    lap(0);
    e->grid_tiles = std::max(1, (int(p.tiles.size()) + kWavesPerBlock - 1) / kWavesPerBlock);
    // one wave past the tiles does the residual bookkeeping -> at least one spare wave
    e->stats.algorithmic_bytes_per_sweep = p.algorithmic_bytes;
    e->stats.layout_bytes_per_sweep = p.layout_bytes;
    e->stats.messages_per_sweep = p.messages_per_sweep;
    if (p.nranks == 1 && !std::getenv("BN_NO_SMALL")) {  // one-workgroup path for small networks (bn_small.hpp)
        try {
            build_small_plan(p, e->small);
        } catch (const std::bad_alloc&) {
            delete e;
            return fail(BN_ERR_ALLOC, "out of host memory while building the small-network plan");
        }
    }
    lap(1);
    if (p.nranks == 1 && !e->small.ok && !std::getenv("BN_NO_MID")) {  // ... spread over several workgroups (bn_mid.hip)
        try {
            build_mid_plan(p, e->mid);
        } catch (const std::bad_alloc&) {
            delete e;
            return fail(BN_ERR_ALLOC, "out of host memory while building the mid-size plan");
        }
    }
    lap(2);
    constexpr int32_t kDagDefaultCap = 224;  // 0.9 x 256 CUs, a multiple of 8 (rebuilt below when the device has another count)
    if (p.nranks == 1 && !std::getenv("BN_NO_DAG")) {  // k = 4, <= 5 parents: register-resident child tiles + parent items (bn_dag.hpp)
        try {
            // LIGHT: the tile tables and the features the default-path policy reads (dag_applies) -- not the padded CPT image, not the
            // device tables: those come with the first use of the path (ensure_dag), at once below where it is the default
            build_dag_plan(p, kDagDefaultCap, e->dag, true);
            e->dag_cap = kDagDefaultCap;
        } catch (const std::bad_alloc&) {   // the other paths can still run the network
            e->dag = DagPlan();
            e->dag.why = "out of host memory while building the plan";
        }
    }
    lap(3);
    if (desc->device == BN_DEVICE_HOST_ONLY) {
        *out = e;
        return BN_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        delete e;
        return fail(BN_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    }
    e->host_only = false;
    DeviceGuard guard;
    const bool dev_timing = std::getenv("BN_CREATE_TIMING") != nullptr;   // where the device side of bn_create goes, one line per step on stderr
    auto dev_t0 = std::chrono::steady_clock::now();
    auto dev_lap = [&](const char* what) {
        if (!dev_timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[bn_create] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - dev_t0).count());
        dev_t0 = now;
    };
    int rc = [&]() -> int {
        if (desc->device >= 0) {
            if (desc->device >= ndev) return fail(BN_ERR_ARG, "device ordinal out of range");
            e->device = desc->device;
        } else {
            HIPCHK(hipGetDevice(&e->device));
        }
        HIPCHK(guard.enter(e->device));
        dev_lap("device + guard");
        e->stream = take_parked_stream(e->device);
        if (!e->stream) HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        dev_lap("stream");
        if (p.nranks > 1) {
            HIPCHK(hipStreamCreateWithFlags(&e->comm_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&e->ev_swept, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&e->ev_gathered, hipEventDisableTiming));
            if (const char* o = std::getenv("BN_OVERLAP")) e->overlap = std::atoi(o) != 0;
        }
        int r;
        if ((r = upload(&e->d_tiles, p.tiles, e->stream))) return r;
        if ((r = upload(&e->d_classes, p.classes, e->stream))) return r;
        if ((r = upload(&e->d_flat_tab, p.flat_tab, e->stream))) return r;
        if ((r = upload(&e->d_cpt, p.cpt_striped, e->stream))) return r;
        if ((r = upload(&e->d_out, p.out_refs, e->stream))) return r;
        if ((r = upload(&e->d_inrefs, p.in_refs, e->stream))) return r;
        if ((r = upload(&e->d_slot_node, p.slot_node, e->stream))) return r;
        if ((r = upload(&e->d_slot_boff, p.slot_boff, e->stream))) return r;
        if ((r = upload(&e->d_node_tile, p.node_tile, e->stream))) return r;
        if ((r = upload(&e->d_node_nl, p.node_nl, e->stream))) return r;
        dev_lap("tile tables + image upload");
        // shards: peers store cut-edge halves straight into these buffers from inside their kernels -> fine-grained
        // (system-coherent) allocations; BN_SHARD_COARSE=1 keeps plain hipMalloc (A/B on one device)
        e->fine_grained = p.nranks > 1 && !std::getenv("BN_SHARD_COARSE");
        for (int i = 0; i < 2; ++i) {
            if (e->fine_grained) {
                HIPCHK(hipExtMallocWithFlags(reinterpret_cast<void**>(&e->d_rec[i]), std::max<size_t>(p.rec_total_doubles, 1) * 8,
                                             hipDeviceMallocFinegrained));
            } else if ((r = dalloc(&e->d_rec[i], size_t(p.rec_total_doubles)))) return r;
            if ((r = dalloc(&e->d_node[i], size_t(p.node_doubles)))) return r;
            HIPCHK(hipMemsetAsync(e->d_rec[i], 0, std::max<size_t>(p.rec_total_doubles, 1) * 8, e->stream));
            HIPCHK(hipMemsetAsync(e->d_node[i], 0, std::max<size_t>(p.node_doubles, 1) * 8, e->stream));
        }
        if ((r = dalloc(&e->d_frozen, size_t(p.n_slots)))) return r;
        HIPCHK(hipMemsetAsync(e->d_frozen, 0, std::max<size_t>(p.n_slots, 1), e->stream));
        // store policy: working sets beyond the Infinity Cache stream their outputs non-temporally
        e->nontemporal = 8 * (p.cpt_doubles + 2 * p.rec_doubles + 2 * p.node_doubles) > (int64_t(192) << 20);
        if ((r = dalloc(&e->d_res_hist, size_t(e->res_cap)))) return r;
        if ((r = dalloc(&e->d_ctl, 1))) return r;
        HIPCHK(hipMemsetAsync(e->d_ctl, 0, sizeof(Ctl), e->stream));  // done_run = 0: no run is marked done
        if ((r = dalloc(&e->d_beliefs, size_t(p.node_off[p.n])))) return r;
        HIPCHK(hipMemsetAsync(e->d_beliefs, 0, std::max<size_t>(p.node_off[p.n], 1) * 8, e->stream));
        dev_lap("state buffers");
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->h_ctl), sizeof(Ctl), hipHostMallocMapped));
        std::memset(e->h_ctl, 0, sizeof(Ctl));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_ctl_dev), e->h_ctl, 0));
        {   // resident path (bn_resident.hip): one-lane tiles (<= 2 parents, <= 8 children per node), one wave per tile, every block co-resident (one 512-thread block of <= 256 VGPRs per CU)
            dev_lap("h_ctl (mapped host)");
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, e->device));
            dev_lap("hipGetDeviceProperties");
            e->n_cus = prop.multiProcessorCount;
            const int64_t nt = int64_t(p.tiles.size());
            // One 8-wave block per CU is two waves per SIMD sharing its issue slots.  A network whose tiles fit the chip at
            // FOUR waves per block (the CPT slots in LDS keep it at one block per CU) gives every wave a SIMD of its own.
            // BN_RESIDENT_WAVES=8 / 4 forces either (A/B).
            e->resident_waves = kResidentWaves;
            {
                // blocks a launch of `w` waves per block needs (single engines round up to a multiple of 8 for the XCD-contiguous
                // mapping, below) + the barrier's service block must fit 0.9 x CUs: decided on the ROUNDED count (a 239 x 240 grid
                // has 898 tiles = 225 blocks of four, 232 after rounding: too many -- it keeps 8 waves per block)
                const int64_t cu_cap = int64_t(prop.multiProcessorCount) * 9 / 10;
                auto fits = [&](int w) {
                    int64_t b = (nt + w - 1) / w;
                    if (b > 1 && p.nranks == 1) b = (b + 7) & ~int64_t(7);
                    return b + 1 <= cu_cap && b <= kResidentMaxBlocks;
                };
                if (nt > kResidentWaves && fits(kResidentWaves / 2)) e->resident_waves = kResidentWaves / 2;
                if (const char* w = std::getenv("BN_RESIDENT_WAVES")) {
                    const int v = std::atoi(w);
                    if (v == kResidentWaves || (v == kResidentWaves / 2 && fits(v)) || (v == 2 && nt > 2 && fits(v))) e->resident_waves = v;
                }
            }
            int64_t nb = (nt + e->resident_waves - 1) / e->resident_waves;
            if (nb > 1 && p.nranks == 1) nb = (nb + 7) & ~int64_t(7);  // XCD-contiguous tile mapping wants a multiple of 8; shards keep the CUs for each other
            bool shapes = (nt == 0 || p.variants == (1 << kVariantUniform)) &&
                          nb + 1 <= int64_t(prop.multiProcessorCount) * 9 / 10 && nb <= kResidentMaxBlocks &&  // + the barrier's service block
                          p.rec_total_doubles * 8 < (int64_t(1) << 31);  // 32-bit byte offsets into a record buffer
            for (const TileDesc& td : p.tiles) shapes = shapes && td.variant == kVariantUniform && td.cmax <= 8 && td.m <= 2;
            bool ok = shapes && p.nranks == 1 && nt > 0;
            for (const TileDesc& td : p.tiles) ok = ok && td.in_ref_base < 0;
            e->shard_shapes_ok = shapes && p.nranks > 1 && p.nranks <= kMaxRanks;
            e->resident_ok = ok;
            e->resident_lean = (ok || e->shard_shapes_ok) && !p.tiles.empty() ? int(p.tiles[0].kv) : 0;
            for (const TileDesc& td : p.tiles)
                if (td.cmax > 2 || int(td.kv) != e->resident_lean) e->resident_lean = 0;
            e->grid_resident = int(nb);
            if (ok) HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->d_rsync), sizeof(ResidentSync)));
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->h_abort), 64, hipHostMallocMapped));
            std::memset(e->h_abort, 0, 64);
            HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_abort_dev), e->h_abort, 0));
            e->flow_ok = ok && nb > 1 && !p.nbr.empty() && nt <= kFlowMaxTiles;
            if (e->flow_ok) {
                HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->d_flow), flow_sync_bytes(1)));
                int r2;
                if ((r2 = upload(&e->d_nbr, p.nbr, e->stream))) return r2;
            }
            if (p.nranks > 1) {
                // shards allocate their page-locked staging now: no allocation call may fall between two ranks' launches of a run
                // (several shard engines in one process: such calls can wait for the whole device)
                const size_t evb = ((size_t(p.n) * 4 + size_t(p.n + 1) * 4 + 7) & ~size_t(7)) + size_t(p.node_off[p.n]) * 8 + 64;
                e->ev_bytes_cap = evb;
                HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->h_ev), evb, hipHostMallocMapped));
                HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_ev_dev), e->h_ev, 0));
                HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->h_beliefs), std::max<size_t>(p.node_off[p.n], 1) * sizeof(double), hipHostMallocMapped));
                HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_beliefs_dev), e->h_beliefs, 0));
            }
            if (e->shard_shapes_ok) {  // zeroed HERE, once: peers write into it from their kernels whenever they run
                if (e->fine_grained)
                    HIPCHK(hipExtMallocWithFlags(reinterpret_cast<void**>(&e->d_flow), flow_sync_bytes(p.nranks), hipDeviceMallocFinegrained));
                else
                    HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->d_flow), flow_sync_bytes(p.nranks)));
                HIPCHK(hipMemsetAsync(e->d_flow, 0, flow_sync_bytes(p.nranks), e->stream));
            }
            if (const char* f = std::getenv("BN_RESIDENT_FLOW")) e->flow = std::atoi(f) != 0;
            if (const char* f = std::getenv("BN_RESIDENT_DIRECT")) e->resident_direct = std::atoi(f) != 0;
            if (const char* f = std::getenv("BN_RESIDENT_DELAY")) e->resident_poll_margin = std::max(-1, std::min(std::atoi(f), 1000));
            if (const char* z = std::getenv("BN_POLL_SLEEP")) e->poll_sleep = std::max(0, std::min(std::atoi(z), 64));
        }
        dev_lap("resident setup");
        {
            if (e->small.ok) {
                const SmallPlan& sp = e->small;
                int r2;
                if ((r2 = upload(&e->d_s_ent, sp.ent, e->stream))) return r2;
                if ((r2 = upload(&e->d_s_cpt, sp.ent_cpt, e->stream))) return r2;
                if ((r2 = upload(&e->d_s_term, sp.term, e->stream))) return r2;
                if ((r2 = upload(&e->d_s_clist, sp.clist, e->stream))) return r2;
                if ((r2 = upload(&e->d_s_bslot, sp.bslot, e->stream))) return r2;
                if ((r2 = upload(&e->d_s_cslot, sp.cslot, e->stream))) return r2;
                if ((r2 = upload(&e->d_s_nvidx, sp.nv_idx, e->stream))) return r2;
                if ((r2 = upload(&e->d_s_nvslot, sp.nv_slot, e->stream))) return r2;
                if ((r2 = upload(&e->d_s_init, sp.npi_init, e->stream))) return r2;
                if ((r2 = dalloc(&e->d_s_state, size_t(2 * sp.M + 2 * sp.N)))) return r2;
                if ((r2 = upload(&e->d_s_nodeoff, sp.node_off, e->stream))) return r2;
                if (int code = prepare_bp_small())
                    return fail(BN_ERR_HIP, std::string("bp_small attribute: ") + hipGetErrorString(hipError_t(code)));
                e->small_ok = true;
            }
        }
        dev_lap("small path");
        if (e->mid.ok && e->n_cus > 0 && int64_t(e->mid.parts.size()) > int64_t(e->n_cus) * 9 / 10) {
            e->mid.ok = false;  // the workgroups of a run wait for each other: one per CU, with room to spare
            e->mid.why = "more workgroups than 0.9 x the device's CUs";
        }
        if (e->mid.ok) {
            const MidPlan& mp = e->mid;
            std::vector<MidPart> parts;
            std::vector<SmallEntry> ent;
            std::vector<double> cpt;
            std::vector<uint32_t> term;
            std::vector<uint16_t> clist;
            std::vector<SmallSlot> bslot, cslot;
            for (const SmallPlan& sp : mp.parts) {
                MidPart pt{sp.v0, sp.v1, int32_t(ent.size()), int32_t(term.size()), int32_t(clist.size()), int32_t(bslot.size()), int32_t(cslot.size()),
                           sp.re, sp.rb, sp.rc, sp.T, sp.TT, sp.CL, sp.waves * kWave};
                parts.push_back(pt);
                ent.insert(ent.end(), sp.ent.begin(), sp.ent.end());
                cpt.insert(cpt.end(), sp.ent_cpt.begin(), sp.ent_cpt.end());
                term.insert(term.end(), sp.term.begin(), sp.term.end());
                clist.insert(clist.end(), sp.clist.begin(), sp.clist.end());
                bslot.insert(bslot.end(), sp.bslot.begin(), sp.bslot.end());
                cslot.insert(cslot.end(), sp.cslot.begin(), sp.cslot.end());
            }
            const SmallPlan& g0 = mp.parts[0];  // (carries the tables over all nodes)
            std::vector<int32_t> msg_first(p.n + 1);
            for (int v = 0; v <= p.n; ++v) msg_first[v] = int32_t(p.msg_off[v < p.n ? p.in_ptr[v] : p.in_ptr[p.n]]);
            int r2;
            if ((r2 = upload(&e->d_m_parts, parts, e->stream))) return r2;
            if ((r2 = upload(&e->d_m_ent, ent, e->stream))) return r2;
            if ((r2 = upload(&e->d_m_cpt, cpt, e->stream))) return r2;
            if ((r2 = upload(&e->d_m_term, term, e->stream))) return r2;
            if ((r2 = upload(&e->d_m_clist, clist, e->stream))) return r2;
            if ((r2 = upload(&e->d_m_bslot, bslot, e->stream))) return r2;
            if ((r2 = upload(&e->d_m_cslot, cslot, e->stream))) return r2;
            if ((r2 = upload(&e->d_m_nvidx, g0.nv_idx, e->stream))) return r2;
            if ((r2 = upload(&e->d_m_nvslot, g0.nv_slot, e->stream))) return r2;
            if ((r2 = upload(&e->d_m_init, g0.npi_init, e->stream))) return r2;
            if ((r2 = upload(&e->d_m_nodeoff, g0.node_off, e->stream))) return r2;
            if ((r2 = upload(&e->d_m_msgfirst, msg_first, e->stream))) return r2;
            if ((r2 = mid_reserve_slots(e, 1))) return r2;
            if (int code = prepare_bp_mid())
                return fail(BN_ERR_HIP, std::string("bp_mid attribute: ") + hipGetErrorString(hipError_t(code)));
            if (const char* mm = std::getenv("BN_MID")) e->mid_mode = std::atoi(mm) != 0;
            e->mid_ok = true;
        }
        dev_lap("mid path");
        if (e->dag.ok) {
            int32_t cap = int32_t((int64_t(e->n_cus) * 9 / 10) & ~int64_t(7));
            if (const char* c = std::getenv("BN_DAG_CAP")) cap = std::max(8, std::min(cap, std::atoi(c) & ~7));   // experiments: fewer blocks
            if (cap != kDagDefaultCap) {
                try {
                    build_dag_plan(p, cap, e->dag, true);
                    e->dag_cap = cap;
                } catch (const std::bad_alloc&) {
                    e->dag = DagPlan();
                    e->dag.why = "out of host memory while building the plan";
                }
            }
        }
        if (e->dag.ok) {
            if (const char* dm = std::getenv("BN_DAG")) e->dag_mode = std::max(0, std::min(2, std::atoi(dm)));
            if (const char* df = std::getenv("BN_DAG_FLOW")) e->dag_flow = std::atoi(df) != 0;
            e->dag_ok = true;   // eligible on this device; image and tables: ensure_dag
        }
        if (const char* m = std::getenv("BN_MULTISWEEP")) e->multisweep = std::max(0, std::min(2, std::atoi(m)));
        lap(4);
        // the register-resident DAG path where the defaults pick it: built and uploaded now (a functor's first query pays nothing);
        // everywhere else with the first run that wants it ("dag" 2, "autotune", a batch)
        if (dag_applies(e) && !std::getenv("BN_LAZY_ALL")) {
            int r2;
            if ((r2 = ensure_dag(e))) return r2;
        }
        lap(3);
        dev_lap("dag path");
        HIPCHK(hipStreamSynchronize(e->stream));
        dev_lap("final synchronise");
        if (p.nranks == 1) prime_spare_streams(e->device);   // (first engine of the process on this device only)
        dev_lap("spare streams");
        BigVec().swap(e->plan.cpt_striped);  // the image now lives in HBM
        lap(4);
        return BN_OK;
    }();
    if (rc != BN_OK) {
        std::string keep = g_err;
        free_engine(e);
        g_err = keep;
        return rc;
    }
    *out = e;
    return BN_OK;
}

extern "C" int bn_create(const bn_model_desc* desc, bn_engine** out) { return create_impl(desc, ShardSpec(), out); }

extern "C" int bn_create_sharded(const bn_model_desc* desc, int32_t rank, int32_t nranks, const int32_t* owner,
                                 bn_engine** out) {
    ShardSpec sh;
    sh.rank = rank;
    sh.nranks = nranks;
    sh.owner = owner;
    return create_impl(desc, sh, out);
}

extern "C" void bn_destroy(bn_engine* eng) { free_engine(eng); }

BpBuffers bn_eng::buffers_of(bn_engine* e) {
    BpBuffers b;
    b.tiles = e->d_tiles;
    b.classes = e->d_classes;
    b.flat_tab = e->d_flat_tab;
    b.n_tiles = int32_t(e->plan.tiles.size());
    b.cpt = e->d_cpt;
    b.rec0 = e->d_rec[0]; b.rec1 = e->d_rec[1];
    b.node0 = e->d_node[0]; b.node1 = e->d_node[1];
    b.out_refs = e->d_out;
    b.frozen = e->d_frozen;
    b.frozen_mark = e->frozen_mark;
    b.slot_node = e->d_slot_node;
    b.slot_boff = e->d_slot_boff;
    b.node_tile = e->d_node_tile;
    b.node_nl = e->d_node_nl;
    b.in_refs = e->d_inrefs;
    b.g_base = e->plan.g_base;
    b.seg_d2 = e->plan.seg_d2;
    b.seg_data_d2 = e->plan.seg_data_d2;
    b.rec_total_doubles = e->plan.rec_total_doubles;
    b.rank = e->plan.rank;
    b.nranks = e->plan.nranks;
    b.res_hist = e->d_res_hist;
    b.res_cap = e->res_cap;
    b.ctl = e->d_ctl;
    b.beliefs = e->beliefs_override ? e->beliefs_override : e->d_beliefs;
    return b;
}

// `seen` / `epoch`: one stamp per node, kept by the engine so that a query costs O(ne), not O(n)
int bn_eng::check_evidence(const Plan& p, int32_t ne, const int32_t* ev_node, const int32_t* ev_off,
                          std::vector<uint32_t>& seen, uint32_t& epoch) {
    if (ne < 0) return fail(BN_ERR_ARG, "negative evidence count");
    if (ne == 0) return BN_OK;
    if (!ev_node || !ev_off) return fail(BN_ERR_ARG, "null evidence array");
    if (ev_off[0] != 0) return fail(BN_ERR_ARG, "ev_off[0] != 0");
    if (seen.size() != size_t(p.n) || epoch == 0xffffffffu) { seen.assign(size_t(p.n), 0u); epoch = 0; }
    ++epoch;
    for (int32_t j = 0; j < ne; ++j) {
        int32_t v = ev_node[j];
        if (v < 0 || v >= p.n) return fail(BN_ERR_ARG, "evidence node out of range");
        if (seen[v] == epoch) return fail(BN_ERR_ARG, "evidence node listed twice");
        seen[v] = epoch;
        if (ev_off[j + 1] - ev_off[j] != p.k[v])
            return fail(BN_ERR_ARG, "evidence vector of node " + std::to_string(v) + " must have selectable_num entries");
    }
    return BN_OK;
}

int bn_eng::ensure_events(bn_engine* e, size_t count) {
    while (e->events.size() < count) {
        hipEvent_t ev;
        HIPCHK(hipEventCreate(&ev));
        e->events.push_back(ev);
    }
    return BN_OK;
}

// Evidence staging: one pinned host block [ev_node | ev_off | ev_val] -> one H2D copy, then the
// evidence is APPLIED (marks cleared, new marks and vectors written): it stays in force for every
// following run until the next call, so a run itself starts with its first sweep.
// wait: block until the upload has left the pinned staging block (bn_bp_set_evidence); bn_bp_run passes false --
// its own single synchronisation at the end of the call covers it
// The evidence in force (staging block) -> the tile buffers: ONE kernel marks the nodes with this set's mark value and writes
// their vectors (bp_evidence_kernel).  No-op when they hold it already.
static int flush_evidence(bn_engine* e) {
    if (!e->ev_deferred) return BN_OK;
    const Plan& p = e->plan;
    if (e->frozen_mark == 255 || e->ev_applied_dirty) {  // the mark values are used up (or a launch failed half-way): start over
        HIPCHK(hipMemsetAsync(e->d_frozen, 0, std::max<size_t>(p.n_slots, 1), e->stream));
        e->frozen_mark = 0;
    }
    ++e->frozen_mark;
    e->ev_applied_dirty = false;
    e->ev_deferred = false;
    EvidenceArgs ea{buffers_of(e), e->ev_ne, e->d_ev_node, e->d_ev_off, e->d_ev_val};
    if (int code = launch_bp_evidence(ea, e->stream)) {
        e->ev_applied_dirty = true;
        return fail(BN_ERR_HIP, std::string("bp_evidence launch failed: ") + hipGetErrorString(hipError_t(code)));
    }
    e->ev_upload_pending = e->ev_ne > 0;
    return BN_OK;
}

static int set_evidence_impl(bn_engine* e, int32_t ne, const int32_t* ev_node, const int32_t* ev_off,
                             const double* ev_val, bool wait) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (e->host_only) return fail(BN_ERR_STATE, "engine was created with BN_DEVICE_HOST_ONLY: no GPU, no compute");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: bn_reload_cpt failed while uploading (destroy it and create a new one)");
    const Plan& p = e->plan;
    int rc = check_evidence(p, ne, ev_node, ev_off, e->ev_seen, e->ev_epoch);
    if (rc) return rc;
    if (ne > 0 && !ev_val) return fail(BN_ERR_ARG, "null ev_val");
    ON_DEVICE(e);
    const int64_t nval = ne > 0 ? ev_off[ne] : 0;
    const size_t off_node = 0, off_off = size_t(ne) * 4, off_val = (off_off + size_t(ne + 1) * 4 + 7) & ~size_t(7);
    const size_t bytes = off_val + size_t(nval) * 8;
    if (e->ev_upload_pending) {  // the staging block is about to be rewritten
        HIPCHK(hipStreamSynchronize(e->stream));
        e->ev_upload_pending = false;
    }
    if (!e->h_ev) {
        // the staging block is sized ONCE, for the largest evidence set the model admits (every node observed): growing it
        // later would mean hipHostFree, which waits for the whole device -- and where several shard engines share a process,
        // a rank whose kernel is already waiting for this rank's would never let that return
        e->ev_bytes_cap = ((size_t(p.n) * 4 + size_t(p.n + 1) * 4 + 7) & ~size_t(7)) + size_t(p.node_off[p.n]) * 8 + 64;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->h_ev), e->ev_bytes_cap, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_ev_dev), e->h_ev, 0));
    }
    if (bytes > e->ev_bytes_cap) return fail(BN_ERR_ARG, "evidence larger than the model");
    // ONE kernel applies a set: it reads the arrays in place from the page-locked staging block (no copy command in the
    // queue in front of the run) and marks the nodes with this set's mark value (no memset of the previous set's marks)
    if (ne > 0) {
        std::memcpy(e->h_ev + off_node, ev_node, size_t(ne) * 4);
        std::memcpy(e->h_ev + off_off, ev_off, size_t(ne + 1) * 4);
        std::memcpy(e->h_ev + off_val, ev_val, size_t(nval) * 8);
    }
    e->d_ev_node = reinterpret_cast<int32_t*>(e->h_ev_dev + off_node);
    e->d_ev_off = reinterpret_cast<int32_t*>(e->h_ev_dev + off_off);
    e->d_ev_val = reinterpret_cast<double*>(e->h_ev_dev + off_val);
    e->ev_ne = ne;
    e->ev_nval = int32_t(nval);
    e->dag_ev_applied = false;
    if (e->small_ok || e->mid_ok || dag_applies(e)) {  // the item kernels read the arrays where they are; flush_evidence() serves every other path
        e->ev_deferred = true;
        e->ev_upload_pending = ne > 0;
        return BN_OK;
    }
    e->ev_deferred = true;
    if ((rc = flush_evidence(e))) return rc;
    if (wait) {
        HIPCHK(hipStreamSynchronize(e->stream));
        e->ev_upload_pending = false;
    }
    return BN_OK;
}

extern "C" int bn_bp_set_evidence(bn_engine* e, int32_t ne, const int32_t* ev_node, const int32_t* ev_off,
                                  const double* ev_val) {
    return set_evidence_impl(e, ne, ev_node, ev_off, ev_val, true);
}

// ---- the steps of a run; bn_bp_run_device chains them, the bn_bp_step_* entry points expose
// them one by one (tests drive several shards on one GPU with an emulated all-gather).
static int step_begin(bn_engine* e) {
    // a run on the tile kernels starts here (also the single-step API): what it leaves behind -- beliefs in d_beliefs, messages
    // in the record buffers -- is what the diagnostics must read, whatever path and output buffer the previous run used
    e->beliefs_on_host_only = false;
    e->last_path = 0;
    if (int rc = flush_evidence(e)) return rc;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    if (!e->rows_clean) {  // the previous run did not end through a finish kernel that saw it over
        if (int code = launch_bp_reset(buffers_of(e), e->stream))
            return fail(BN_ERR_HIP, std::string("bp_reset launch failed: ") + hipGetErrorString(hipError_t(code)));
    }
    e->rows_clean = false;
    return BN_OK;
}

// part: 0 = every tile + bookkeeping (one launch); 1 = interior tiles only, no bookkeeping;
// 2 = the tiles that touch a cut edge + bookkeeping (sharded runs, see bn_plan.cpp / run loop)
static int step_sweep(bn_engine* e, int32_t sweep, double eps, int part = 0) {
    const int cur = sweep & 1;
    const int32_t nt = int32_t(e->plan.tiles.size()), ni = e->plan.n_interior_tiles;
    const int32_t t0 = part == 2 ? ni : 0, t1 = part == 1 ? ni : nt;
    const int32_t book = part == 1 ? 0 : 1;
    if (t1 - t0 + book <= 0) return BN_OK;
    SweepArgs sa{buffers_of(e), e->d_rec[cur], e->d_rec[cur ^ 1], e->d_node[cur], e->d_node[cur ^ 1], eps, sweep,
                 t0, t1, book, e->run_id, SetStrides{}};
    // one wave per tile (+ one for the bookkeeping), blocks padded to a multiple of 8 for the XCD mapping
    const int grid = ((t1 - t0 + book + kWavesPerBlock - 1) / kWavesPerBlock + 7) & ~7;
    static const bool no_light = std::getenv("BN_NO_LIGHT") != nullptr;  // A/B switch
    if (launch_bp_sweep(sa, grid, 1, e->nontemporal, e->plan.light && !no_light, e->plan.variants, e->stream))
        return fail(BN_ERR_HIP, "bp_sweep launch failed");
    return BN_OK;
}

// Halo exchange after sweep `sweep`: in-place all-gather of every rank's segment of the buffer
// that sweep wrote.  One collective per sweep; it also carries the residual slots.
static int step_exchange(bn_engine* e, int32_t sweep, hipStream_t on) {
    const Plan& p = e->plan;
    // BN_EXCHANGE_ALWAYS: issue the (then trivial) collective on a 1-rank communicator too, so the
    // RCCL call can be exercised on a single-GPU box
    if (p.nranks == 1 && !(e->comm && std::getenv("BN_EXCHANGE_ALWAYS"))) return BN_OK;
    if (!e->comm) return fail(BN_ERR_COMM, "sharded engine: call bn_comm_init before running");
    double* g = e->d_rec[(sweep + 1) & 1] + 2 * p.g_base;
    const size_t count = size_t(2 * p.seg_d2);
    ncclResult_t rc = g_rccl.AllGather(g + size_t(p.rank) * count, g, count, ncclDouble, e->comm, on);
    if (rc != ncclSuccess) return fail(BN_ERR_COMM, std::string("ncclAllGather: ") + g_rccl.GetErrorString(rc));
    return BN_OK;
}

// One iteration of a sharded run with the exchange overlapped: the interior tiles of
// iteration s read nothing the all-gather of iteration s-1 delivers, so their launch goes out first and
// runs while that collective is still in flight on the comm stream; only the launch over the tiles that
// touch a cut edge (and the residual bookkeeping, which reads every rank's slots) waits for it.
// Critical path per iteration: max(interior kernel, all-gather) + boundary kernel, instead of their sum.
static int step_sweep_overlapped(bn_engine* e, int32_t sweep, double eps, bool gather_pending) {
    int rc;
    if ((rc = step_sweep(e, sweep, eps, 1))) return rc;
    if (gather_pending) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_gathered, 0));
    if ((rc = step_sweep(e, sweep, eps, 2))) return rc;
    HIPCHK(hipEventRecord(e->ev_swept, e->stream));
    HIPCHK(hipStreamWaitEvent(e->comm_stream, e->ev_swept, 0));
    if ((rc = step_exchange(e, sweep, e->comm_stream))) return rc;
    HIPCHK(hipEventRecord(e->ev_gathered, e->comm_stream));
    return BN_OK;
}

static int step_finish(bn_engine* e, int32_t launched, bool final_batch, double eps) {
    // wave 0 writes the outcome straight into the pinned host Ctl: visible after the stream
    // synchronises, no copy command in between
    FinishArgs fa{buffers_of(e), eps, launched, final_batch ? 1 : 0, e->run_id, e->h_ctl_dev, SetStrides{}};
    if (launch_bp_finish(fa, e->grid_tiles, 1, e->stream)) return fail(BN_ERR_HIP, "bp_finish launch failed");
    return BN_OK;
}

static void note_run_result(bn_engine* e) {
    e->rows_clean = true;  // the finish kernel that saw the run over left the residual slots zero
    e->last_ctl = *e->h_ctl;
    e->have_run = true;
    e->predicted_sweeps = e->last_ctl.n_sweeps;
    e->stats.sweeps = e->last_ctl.n_sweeps;
    // device clock (100 MHz): start of sweep 0 -> start of the launch after the last executed sweep
    const unsigned long long t0 = e->last_ctl.t_first, t1 = e->last_ctl.t_last;
    e->stats.sweep_devclock_ms = t1 > t0 ? float(double(t1 - t0) * 1e-5) : 0.f;
}

// blocks the barrier of a resident launch adds to the tile blocks: one, sweeping every tile block's granules
int bn_eng::resident_service_blocks(int tile_blocks) { return tile_blocks > 1 ? 1 : 0; }

// Networks of register-resident tiles that fit the chip: ONE launch runs the whole run with the CPTs,
// references and node vectors resident in registers / LDS and a grid barrier per sweep (bn_resident.hip).
// BN_ERR_STATE = a bounded wait inside the kernel gave up: the caller redoes the run with per-sweep launches.
// copy_to: host memory the beliefs are copied into BEHIND the launch, before the run's one synchronisation
// (bn_bp_run / bn_bp_run_view); nullptr leaves them in HBM (bn_bp_run_device)
static int run_resident(bn_engine* e, double eps, int32_t max_sweeps, double* copy_to) {
    hipStream_t s = e->stream;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    int32_t begin = 0, launches = 0;
    float ms = 0.f;
    double dev_ticks = 0.0;
    for (;;) {
        // polled words: generations count on from launch to launch, so they are zeroed only at creation, after an
        // aborted launch and before the 30-bit generation would wrap
        const bool shard = e->plan.nranks > 1;  // (only called with shard_flow_ok then)
        const bool flow = shard || (e->flow_ok && e->flow != 0);
        if (shard) {
            // Every rank derives the generations of a launch from the number of runs the engine has been asked for and
            // the launch's place in the run: ranks agree without talking, nothing is ever zeroed while peers may be
            // writing, and a granule left by an earlier (or an aborted) launch can never carry a wanted generation.
            if (launches >= 4) return fail(BN_ERR_STATE, "sharded resident run needs more than 4 launches");
            e->flow_gen_base = (((e->shard_run_seq & 0x3ffffu) << 2) + uint32_t(launches)) * uint32_t(kResidentBudget + 1);
        } else if (flow) {
            if (e->flow_dirty || e->flow_gen_base > (1u << 29)) {
                HIPCHK(hipMemsetAsync(e->d_flow, 0, flow_sync_bytes(1), s));
                e->flow_dirty = false;
                e->flow_gen_base = 0;
            }
        } else if (e->rsync_dirty || e->gen_base > (1u << 29)) {
            HIPCHK(hipMemsetAsync(e->d_rsync, 0, sizeof(ResidentSync), s));
            e->rsync_dirty = false;
            e->gen_base = 0;
        }
        *e->h_abort = 0;
        ResidentArgs a{buffers_of(e), eps, max_sweeps, begin, kResidentBudget, e->run_id, flow ? e->flow_gen_base : e->gen_base,
                       // one wait: 50 ms of the 100 MHz clock; shards: 2 s (the ranks' launches start up to a host hiccup apart)
                       shard ? 200000000ull : 5000000ull, e->d_rsync, e->h_ctl_dev,
                       e->grid_resident, e->resident_waves, 1, 1u, 0, 0, 0, 0, 0, flow ? e->d_flow : nullptr,
                       shard ? e->d_peers : nullptr, shard ? e->d_pub_mask : nullptr, shard ? e->plan.n_interior_tiles : 0,
                       e->d_nbr, e->plan.nbr_chunks, e->poll_sleep, e->h_abort_dev, (!flow && !shard) ? e->resident_direct : 0, e->resident_poll_margin};
        if (e->timing) {
            int rc = ensure_events(e, 2);
            if (rc) return rc;
            HIPCHK(hipEventRecord(e->events[0], s));
        }
        if (int code = launch_bp_resident(a, e->grid_resident + (shard ? 1 : resident_service_blocks(e->grid_resident)), e->resident_lean, s))
            return fail(BN_ERR_HIP, std::string("bp_resident launch failed: ") + hipGetErrorString(hipError_t(code)));
        if (e->timing) HIPCHK(hipEventRecord(e->events[1], s));
        // (shards: the copy goes out only once the kernel has ended -- a copy into pageable memory blocks inside the runtime,
        // and where several shard engines live in one process, the thread of a rank whose kernel is still waiting for a
        // peer's would keep that peer's thread from launching)
        if (copy_to && !shard)  // a launch that stops on its budget (1024 sweeps) copies an intermediate state; the last one counts
            HIPCHK(hipMemcpyAsync(copy_to, e->d_beliefs, sizeof(double) * e->plan.node_off[e->plan.n], hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (copy_to && shard) {  // through the engine's page-locked buffer: a plain DMA, nothing that blocks inside the runtime
            const size_t bytes = sizeof(double) * e->plan.node_off[e->plan.n];
            if (copy_to != e->h_beliefs && !e->h_beliefs) {
                HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->h_beliefs), std::max<size_t>(bytes, 8), hipHostMallocMapped));
                HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_beliefs_dev), e->h_beliefs, 0));
            }
            HIPCHK(hipMemcpyAsync(e->h_beliefs, e->d_beliefs, bytes, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (copy_to != e->h_beliefs) std::memcpy(copy_to, e->h_beliefs, bytes);
        }
        e->ev_upload_pending = false;
        ++launches;
        if (!shard) (flow ? e->flow_gen_base : e->gen_base) += kResidentBudget + 1;
        const bool gave_up = e->h_ctl->done < 0 || *e->h_abort != 0;  // any block may raise it, whatever block 0 / the service reported
        if (e->h_ctl->run_id != e->run_id || gave_up) (flow ? e->flow_dirty : e->rsync_dirty) = true;
        if (gave_up) {
            char where[96];
            std::snprintf(where, sizeof where, " (code 0x%x: wait kind %u, iteration %u, tile %u; run seq %u)", *e->h_abort, *e->h_abort & 0xffu,
                          (*e->h_abort >> 8) & 0xfffu, (*e->h_abort >> 20) & 0x7ffu, e->shard_run_seq);
            return fail(BN_ERR_STATE, std::string("resident kernel gave up a bounded wait") + where);
        }
        if (e->h_ctl->run_id != e->run_id) return fail(BN_ERR_STATE, "resident kernel did not report (stale control block)");
        if (e->timing) {
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, e->events[0], e->events[1]));
            ms += t;
        }
        dev_ticks += double(e->h_ctl->t_last - e->h_ctl->t_first);
        if (e->h_ctl->done != 0) break;
        begin = e->h_ctl->n_sweeps;
    }
    const bool rows_were_clean = e->rows_clean;  // this path never touches the residual slots
    note_run_result(e);
    e->rows_clean = rows_were_clean;
    e->last_path = 2;
    e->last_flow = (e->plan.nranks > 1 || (e->flow_ok && e->flow != 0)) ? 1 : 0;
    e->stats.sweep_launches = launches;
    e->stats.sweep_kernel_ms = ms;
    e->stats.sweep_devclock_ms = float(dev_ticks * 1e-5);
    return BN_OK;
}

SmallArgs bn_eng::small_args_of(bn_engine* e, const BpBuffers& b, double eps, int32_t max_sweeps, int32_t begin, Ctl* host_ctl) {
    const SmallPlan& sp = e->small;
    SmallArgs a{};
    a.b = b; a.eps = eps; a.max_sweeps = max_sweeps; a.sweep_begin = begin; a.budget = kSmallBudget; a.run_id = e->run_id;
    a.host_ctl = host_ctl;
    a.n = sp.n; a.N = sp.N; a.M = sp.M; a.S = sp.S; a.T = sp.T; a.TT = sp.TT; a.CL = sp.CL;
    a.re = sp.re; a.rb = sp.rb; a.rc = sp.rc; a.mmax = sp.mmax;
    a.ent = e->d_s_ent; a.ent_cpt = e->d_s_cpt; a.term = e->d_s_term; a.clist = e->d_s_clist;
    a.bslot = e->d_s_bslot; a.cslot = e->d_s_cslot; a.nv_idx = e->d_s_nvidx; a.nv_slot = e->d_s_nvslot; a.npi_init = e->d_s_init;
    a.state = e->d_s_state; a.sets = SetStrides{}; a.state_stride = 0;
    a.ev_mode = 0; a.ev_ne = 0; a.ev_nval = 0; a.ev_node = nullptr; a.ev_off = nullptr; a.ev_val = nullptr; a.ev_meta = nullptr;
    a.node_off = e->d_s_nodeoff;
    return a;
}

// Small networks: ONE workgroup runs every iteration with the state in LDS and writes the beliefs (bn_small.hip).
static int run_small(bn_engine* e, double eps, int32_t max_sweeps, double* copy_to) {
    hipStream_t s = e->stream;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    int32_t begin = 0, launches = 0;
    float ms = 0.f;
    double dev_ticks = 0.0;
    for (;;) {
        SmallArgs a = small_args_of(e, buffers_of(e), eps, max_sweeps, begin, e->h_ctl_dev);
        if (e->ev_deferred) {  // the evidence in force was never written to the tile buffers: the kernel reads the staging block
            a.ev_mode = 1; a.ev_ne = e->ev_ne; a.ev_nval = e->ev_nval; a.ev_node = e->d_ev_node; a.ev_off = e->d_ev_off; a.ev_val = e->d_ev_val;
        }
        if (e->timing) {
            int rc = ensure_events(e, 2);
            if (rc) return rc;
            HIPCHK(hipEventRecord(e->events[0], s));
        }
        if (int code = launch_bp_small(a, e->small.waves, e->small.lds_bytes, 1, s))
            return fail(BN_ERR_HIP, std::string("bp_small launch failed: ") + hipGetErrorString(hipError_t(code)));
        if (e->timing) HIPCHK(hipEventRecord(e->events[1], s));
        if (copy_to)  // a launch that stops on its budget copies an intermediate state; the last one counts
            HIPCHK(hipMemcpyAsync(copy_to, e->d_beliefs, sizeof(double) * e->plan.node_off[e->plan.n], hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        e->ev_upload_pending = false;
        ++launches;
        if (e->h_ctl->run_id != e->run_id) return fail(BN_ERR_HIP, "bp_small kernel did not report (stale control block)");
        if (e->timing) {
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, e->events[0], e->events[1]));
            ms += t;
        }
        dev_ticks += double(e->h_ctl->t_last - e->h_ctl->t_first);
        if (e->h_ctl->done != 0) break;
        begin = e->h_ctl->n_sweeps;
    }
    const bool rows_were_clean = e->rows_clean;  // this path never touches the residual slots
    note_run_result(e);
    e->rows_clean = rows_were_clean;
    e->last_path = 3;
    e->stats.sweep_launches = launches;
    e->stats.sweep_kernel_ms = ms;
    e->stats.sweep_devclock_ms = float(dev_ticks * 1e-5);
    return BN_OK;
}

// the mid-size kernel is the path of choice for this engine (measured: grids, chains and trees run faster on the resident tiles)
bool bn_eng::mid_applies(const bn_engine* e) {
    if (!e->mid_ok || e->multisweep == 0 || e->mid_mode == 0) return false;
    if (e->mid_mode == 2 || !e->resident_ok) return true;
    // Networks the resident tiles cover as well (scripts/experiments/mid_path.py, us per sweep resident / this path): with two
    // parents per node and four states the tile's 64-entry contraction costs more than the items (16 x 16 grid 5.9 / 4.3,
    // 32 x 32 7.0 / 6.4, 200-node DAG 6.7 / 4.3); chains, trees and smaller tables stay on the tiles (400-node chain 3.2 / 4.3,
    // 12 x 12 grid of k = 3: 3.7 / 4.3).
    int kmax = 0;
    for (int32_t k : e->plan.k) kmax = std::max(kmax, int(k));
    // ... up to the size of the 40 x 40 grid (6.6 against 6.6-6.9); larger ones stay on the tiles (64 x 64: 6.4)
    int mmax = 0;  // (over all parts: the first one may hold a grid's first row only)
    for (const SmallPlan& sp : e->mid.parts) mmax = std::max(mmax, int(sp.mmax));
    return mmax >= 2 && kmax >= 4 && e->mid.est_total <= 580000;
}

// Networks spread over several workgroups (bn_mid.hip).  The arguments of a launch over the sets [set_base, set_base + n)
// of a batch (single query: set 0 of one) working in state slots [0, n).
MidArgs bn_eng::mid_args_of(bn_engine* e, const BpBuffers& b0, const SetStrides& st, Ctl* h_ctl_dev, double eps, int32_t max_sweeps,
                           int32_t begin, int32_t set_base, int32_t slot_base) {
    const SmallPlan& g0 = e->mid.parts[0];
    MidArgs a{};
    a.b = b0; a.eps = eps; a.max_sweeps = max_sweeps; a.sweep_begin = begin; a.budget = kSmallBudget; a.run_id = e->run_id;
    a.host_ctl = h_ctl_dev;
    a.n = g0.n; a.N = g0.N; a.M = g0.M; a.nparts = int32_t(e->mid.parts.size());
    a.parts = e->d_m_parts; a.ent = e->d_m_ent; a.ent_cpt = e->d_m_cpt; a.term = e->d_m_term; a.clist = e->d_m_clist;
    a.bslot = e->d_m_bslot; a.cslot = e->d_m_cslot; a.nv_idx = e->d_m_nvidx; a.nv_slot = e->d_m_nvslot; a.npi_init = e->d_m_init;
    a.node_off = e->d_m_nodeoff; a.msg_first = e->d_m_msgfirst;
    a.ev_mode = 0; a.ev_ne = 0; a.ev_node = nullptr; a.ev_off = nullptr; a.ev_val = nullptr; a.ev_meta = nullptr;
    a.state_stride = 4 * int64_t(g0.M) + 4 * int64_t(g0.N);
    a.pi = e->d_m_state; a.lam = a.pi + 2 * size_t(g0.M); a.npi = a.lam + 2 * size_t(g0.M); a.nlam = a.npi + 2 * size_t(g0.N);
    a.frz = e->d_m_frz;
    a.bar = reinterpret_cast<unsigned*>(e->d_m_sync);
    a.res = reinterpret_cast<unsigned long long*>(e->d_m_sync + 8);
    a.abort = e->h_abort_dev;
    a.timeout_ticks = 5000000ull;  // one wait: 50 ms of the 100 MHz clock
    // first poll of the grid barrier placed by the previous barrier's lag (arrival times in the granules, as bn_dag.hip / bn_resident.hip do):
    // BN_MID_DELAY = margin in 10 ns ticks, -1 (default) = poll from the own arrival on.  mixed10k, us per sweep: 8.30 off, 8.17 at 0,
    // 8.80 at 30, 9.08 at 60 (round 6; round 5 measured 8.5 / 8.7 / 9.1): the polling wave has nothing else to do and polls back to
    // back, so a poll placed by prediction can only be later -- at margin 0 it is within the run-to-run spread, with any margin slower
    static const int mid_first_delay = std::getenv("BN_MID_DELAY") ? std::atoi(std::getenv("BN_MID_DELAY")) : -1;
    a.first_poll_delay = mid_first_delay;
    a.sets = st; a.set_base = set_base; a.slot_base = slot_base;
    return a;
}
// launch + wait; BN_ERR_STATE: a grid wait gave up (the caller redoes the work on the tile kernels)
// wait = false: enqueue only (the chunks of a batch, bn_engine_batch.cpp: the caller clears the abort word before the first, waits once
// behind the last and looks at the abort word then)
int bn_eng::mid_launch(bn_engine* e, const MidArgs& a, int32_t n_sets, const double* copy_from, double* copy_to, bool wait) {
    hipStream_t s = e->stream;
    if (wait) *e->h_abort = 0;
    HIPCHK(hipMemsetAsync(e->d_m_sync + size_t(a.slot_base) * kMidSyncBytes, 0, size_t(n_sets) * kMidSyncBytes, s));
    if (int code = launch_bp_mid(a, e->mid.waves, e->mid.rounds, e->mid.lds_bytes, n_sets, s))
        return fail(BN_ERR_HIP, std::string("bp_mid launch failed: ") + hipGetErrorString(hipError_t(code)));
    if (copy_to) HIPCHK(hipMemcpyAsync(copy_to, copy_from, sizeof(double) * e->plan.node_off[e->plan.n], hipMemcpyDeviceToHost, s));
    if (!wait) return BN_OK;
    HIPCHK(hipStreamSynchronize(s));
    e->ev_upload_pending = false;
    if (*e->h_abort != 0) {
        *e->h_abort = 0;
        return fail(BN_ERR_STATE, "a workgroup of the mid-size kernel gave up its grid wait");
    }
    return BN_OK;
}
// one query: one launch for the whole run (more only beyond 65 536 iterations)
static int run_mid(bn_engine* e, double eps, int32_t max_sweeps, double* copy_to) {
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    int32_t begin = 0, launches = 0;
    double dev_ticks = 0.0;
    const BpBuffers b = buffers_of(e);
    for (;;) {
        MidArgs a = mid_args_of(e, b, SetStrides{}, e->h_ctl_dev, eps, max_sweeps, begin, 0, 0);
        if (e->ev_deferred) {  // the evidence in force was never written to the tile buffers: the kernel reads the staging block
            a.ev_mode = 1; a.ev_ne = e->ev_ne; a.ev_node = e->d_ev_node; a.ev_off = e->d_ev_off; a.ev_val = e->d_ev_val;
        }
        if (int rc = mid_launch(e, a, 1, b.beliefs, copy_to)) return rc;
        ++launches;
        if (e->h_ctl->done < 0) return fail(BN_ERR_STATE, "a workgroup of the mid-size kernel gave up its grid wait");
        if (e->h_ctl->run_id != e->run_id) return fail(BN_ERR_HIP, "bp_mid kernel did not report (stale control block)");
        dev_ticks += double(e->h_ctl->t_last - e->h_ctl->t_first);
        if (e->h_ctl->done != 0) break;
        begin = e->h_ctl->n_sweeps;
    }
    const bool rows_were_clean = e->rows_clean;  // this path never touches the residual slots
    note_run_result(e);
    e->rows_clean = rows_were_clean;
    e->last_path = 4;
    e->stats.sweep_launches = launches;
    e->stats.sweep_kernel_ms = 0.f;
    e->stats.sweep_devclock_ms = float(dev_ticks * 1e-5);
    return BN_OK;
}

// k = 4 networks with up to 5 parents per node whose size puts them beyond the item kernels (BASELINE configs[1]): the
// register-resident DAG path (bn_dag.hip) where no other one-launch path takes the network; "dag" 2 = wherever eligible
bool bn_eng::dag_applies(const bn_engine* e) {
    if (!e->dag_ok || e->multisweep == 0 || e->dag_mode == 0) return false;
    if (e->dag_mode == 2) return true;
    // Networks the one-workgroup path (state in LDS) takes as well, us per query (profiles/r05_paths.json), that path / this one: one
    // round of entry items (ALARM-sized) 48.6 / 68.7, Pearl's four nodes 19.3 / 24.5 -- but 8 x 8 grid, k = 4 (four rounds) 86.0 / 64.4,
    // 60 nodes of mixed arity with <= 3 parents (three rounds) 72.3 / 64.6.  Chains and trees the resident tiles run in ONE block
    // stay there (200-node chain: 71.9 resident, 81.6 this path, 113.6 one workgroup).
    if (e->small_ok && e->small.re <= 2) return false;   // (two rounds: not measured; the one-workgroup path also keeps the reference's order for >= 3 parents)
    // ... and where the two paths' BITS differ -- some node has >= 3 parents: lane groups re-associate, the one-workgroup path keeps the
    // reference's order -- the small network stays on the one-workgroup path whatever its rounds: a batch of such a network runs one
    // workgroup per set (bn_engine_batch.cpp), and a set's answer must not depend on whether it was asked alone or in a batch
    // (scripts/soak_gpu.py found a 30-node network where the two differed by 2e-16; round 6).  Price: 60 nodes of mixed arity, <= 3
    // parents: 72 instead of 65 us per query.
    if (e->small_ok && e->dag.has_groups) return false;
    if (e->small_ok && e->small.mmax <= 1 && e->resident_ok && e->grid_resident == 1) return false;
    // Arities below 4 (padded form), us per sweep, this path / the default before (scripts/time_dag_mixed.py): mixed arities 2-4 with
    // <= 3 parents 300 / 3 000 / 10 000 nodes 4.6 / 5.4, 5.7 / 7.1, 6.5 / 9.2 (item kernels); <= 4 parents, 10 000 nodes (723 k entries:
    // beyond the item kernels) 6.5 / 32.5; binary, <= 4 parents, 10 000 nodes 6.2 / 7.7; k = 3 grid 64 x 64 5.0 / 6.1 -- but k = 2 grid
    // 128 x 128 6.4 / 5.5 (resident tiles): an eighth of every padded table is real there.
    if (!e->dag.uniform4 && !e->dag.has_groups && e->dag.fill < 0.25) return false;
    // Measured, us per query (evidence staged, profiles/r04_paths.json), this path / the best of the others:
    //   lane-group tiles (some node has 3-5 parents): 200 nodes 76 / 122, 1 000 nodes 87 / 165, 3 000 nodes 101 / 182, 10 000 nodes
    //   (BASELINE configs[1]) 117 / 215; nodes of <= 2 parents: 16 x 16 grid 86 / 95, 40 x 40 117 / 135, 64 x 64 113 / 139, 128 x 128
    //   133 / 146, 3 000-node DAG 106 / 119, 200-node chain 74 / 74 -- but 200 x 200 grid 283 / 151, 316 x 316 634 / 234: there the
    //   network no longer fits the chip at one tile per wave (stream form) and the resident tiles keep it.
    // stream form re-reads the padded image every sweep: not where less than a quarter of it is real (a padded binary network
    // with 5-parent nodes is 64x its model), whatever the parent counts -- only <= 10 k-node networks were measured in that form
    if (e->dag.stream && e->dag.fill < 0.25) return false;
    if (e->dag.has_groups) return true;
    return !e->dag.stream;
}

// The evidence in force (staging block) -> the state arrays of the DAG path: marks of this set's own value, vectors in both buffers.
static int flush_dag_evidence(bn_engine* e) {
    if (int rc = ensure_dag(e)) return rc;
    if (e->dag_ev_applied) return BN_OK;
    if (e->dag_mark == 255) {  // the mark values are used up: start over
        HIPCHK(hipMemsetAsync(e->d_g_frz, 0, size_t(e->dag.n), e->stream));
        e->dag_mark = 0;
    }
    ++e->dag_mark;
    DagEvidenceArgs ea{e->ev_ne, e->dag.n, e->dag.E, e->d_ev_node, e->d_ev_off, e->d_ev_val, e->d_g_state, e->d_g_frz, e->dag_mark, e->d_g_k, e->d_g_nperm};
    if (int code = launch_dag_evidence(ea, e->stream))
        return fail(BN_ERR_HIP, std::string("dag_evidence launch failed: ") + hipGetErrorString(hipError_t(code)));
    e->dag_ev_applied = true;
    e->ev_upload_pending = e->ev_ne > 0;
    return BN_OK;
}

// One launch runs the whole query (more only beyond kDagBudget iterations).  BN_ERR_STATE: a grid wait gave up.
int bn_eng::run_dag(bn_engine* e, double eps, int32_t max_sweeps, double* copy_to) {
    hipStream_t s = e->stream;
    if (int rc = ensure_dag(e)) return rc;   // (first use of the path on this engine: full plan, device tables, upload)
    const DagPlan& dp = e->dag;
    if (int rc = flush_dag_evidence(e)) return rc;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    int32_t begin = 0, launches = 0;
    float ms = 0.f;
    double dev_ticks = 0.0;
    const BpBuffers b = buffers_of(e);
    if (!dp.uniform4) {   // arities below 4: the run's initial state stands in memory (zeros in the padding), bn_dag_plan.cpp
        DagInitArgs ia{dp.n, dp.E, e->d_g_inptr, e->d_g_inidx, e->d_g_k, e->d_g_init, e->d_g_state, e->d_g_frz, e->dag_mark, e->d_g_eperm, e->d_g_nperm};
        if (int code = launch_dag_init(ia, s))
            return fail(BN_ERR_HIP, std::string("dag_init launch failed: ") + hipGetErrorString(hipError_t(code)));
    }
    for (;;) {
        // polled words: generations count on from launch to launch; zeroed at creation, after an abort and before they would wrap
        if (e->dag_sync_dirty || e->dag_gen_base > (1u << 29)) {
            HIPCHK(hipMemsetAsync(e->d_g_sync, 0, sizeof(ResidentSync), s));
            if (e->d_g_flow) HIPCHK(hipMemsetAsync(e->d_g_flow, 0, dag_flow_sync_bytes(dp.tiles.size()), s));
            e->dag_sync_dirty = false;
            e->dag_gen_base = 0;
        }
        *e->h_abort = 0;
        DagArgs a{};
        // the dataflow form where the plan allows it ("dagflow" 1; a run that gave up a wait stays on the barrier for a while)
        const bool flow = e->dag_flow_ok && e->dag_flow != 0 && !dp.stream && dp.blocks > 1 && e->dag_flow_pause == 0;
        if (flow) {
            static const int flow_sleep = std::getenv("BN_DAG_FLOW_SLEEP") ? std::atoi(std::getenv("BN_DAG_FLOW_SLEEP")) : 4;   // x 512 cycles between polls; configs[1], us per executed iteration: 6.59 / 6.38 / 6.14 / 6.06 / 6.00 at 0 / 1 / 2 / 4 / 8
            a.flow = e->d_g_flow; a.nbr = e->d_g_nbr; a.n_tiles = int32_t(dp.tiles.size()); a.flow_sleep = flow_sleep;
        }
        e->last_dag_flow = flow ? 1 : 0;
        a.b = b; a.eps = eps; a.max_sweeps = max_sweeps; a.sweep_begin = begin; a.budget = kDagBudget; a.run_id = e->run_id;
        a.gen_base = e->dag_gen_base;
        a.timeout_ticks = 5000000ull;  // one wait: 50 ms of the 100 MHz clock
        a.sync = e->d_g_sync; a.host_ctl = e->h_ctl_dev; a.host_abort = e->h_abort_dev;
        a.n = dp.n; a.E = dp.E; a.n_blocks = dp.blocks;
        a.tiles = e->d_g_tiles; a.slot_ptr = e->d_g_slotptr; a.cnode = e->d_g_cnode; a.pitem = e->d_g_pitem; a.oedge = e->d_g_oedge;
        a.cpt_img = e->d_g_cpt; a.npi_init = e->d_g_init; a.state = e->d_g_state; a.frz = e->d_g_frz; a.frz_mark = e->dag_mark;
        static const int poll_sleep = std::getenv("BN_DAG_SLEEP") ? std::atoi(std::getenv("BN_DAG_SLEEP")) : 1;
        a.poll_sleep = poll_sleep;
        static const int first_delay = std::getenv("BN_DAG_DELAY") ? std::atoi(std::getenv("BN_DAG_DELAY")) : 30;   // 10 ns ticks: measured flat from 20 to 60 (config 2: 6.9 us per sweep at 0, 6.5-6.6 there)
        a.first_poll_delay = first_delay;
        a.n_sets = 1; a.set_mask = 1u;
        a.state_init = dp.uniform4 ? 0 : 1; a.node_k = e->d_g_k; a.node_off = e->d_g_noff;
        if (e->timing) {
            int rc = ensure_events(e, 2);
            if (rc) return rc;
            HIPCHK(hipEventRecord(e->events[0], s));
        }
        if (int code = launch_bp_dag(a, dp.stream, s))
            return fail(BN_ERR_HIP, std::string("bp_dag launch failed: ") + hipGetErrorString(hipError_t(code)));
        if (e->timing) HIPCHK(hipEventRecord(e->events[1], s));
        if (copy_to)  // a launch that stops on its budget copies an intermediate state; the last one counts
            HIPCHK(hipMemcpyAsync(copy_to, e->d_beliefs, sizeof(double) * e->plan.node_off[e->plan.n], hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        e->ev_upload_pending = false;
        ++launches;
        e->dag_gen_base += kDagBudget + 1;
        const bool gave_up = e->h_ctl->done < 0 || *e->h_abort != 0;
        if (e->h_ctl->run_id != e->run_id || gave_up) e->dag_sync_dirty = true;
        if (gave_up) {
            *e->h_abort = 0;
            if (flow) e->dag_flow_pause = 64;   // (the barrier form next time the path is tried)
            return fail(BN_ERR_STATE, "a block of the register-resident DAG kernel gave up its grid wait");
        }
        if (e->h_ctl->run_id != e->run_id) return fail(BN_ERR_HIP, "bp_dag kernel did not report (stale control block)");
        if (!flow && e->dag_flow_pause > 0) --e->dag_flow_pause;
        if (e->timing) {
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, e->events[0], e->events[1]));
            ms += t;
        }
        dev_ticks += double(e->h_ctl->t_last - e->h_ctl->t_first);
        if (e->h_ctl->done != 0) break;
        begin = e->h_ctl->n_sweeps;
    }
    const bool rows_were_clean = e->rows_clean;  // this path never touches the residual slots
    note_run_result(e);
    e->rows_clean = rows_were_clean;
    e->last_path = 5;
    e->stats.sweep_launches = launches;
    e->stats.sweep_kernel_ms = ms;
    e->stats.sweep_devclock_ms = float(dev_ticks * 1e-5);
    return BN_OK;
}

// A one-launch path gave up a bounded wait: its workgroups were not all on the chip together -- another engine, stream or process
// holds compute units.  The run is repeated on a slower path and the result is the same, but the caller should know why its
// queries got slower: ONE line per engine on stderr (not gated by BN_DEBUG); the counters keep counting
// (bn_bp_stats.resident_aborts, bn_get_info "mid_aborts" / "dag_aborts").
void bn_eng::report_abort_once(bn_engine* e, const char* what, int pause_runs) {
    if (e->abort_reported && !std::getenv("BN_DEBUG")) return;
    e->abort_reported = true;
    std::fprintf(stderr,
                 "[bn_mi355x] %s gave up a bounded wait (%s): its workgroups were not all resident -- does another engine, stream or "
                 "process use this GPU?  This run and the next %d take a slower path (same results); further such events are counted, "
                 "not printed (bn_bp_stats.resident_aborts, bn_get_info \"mid_aborts\" / \"dag_aborts\").\n",
                 what, g_err.c_str(), pause_runs);
}

static int run_device_impl(bn_engine* e, double eps, int32_t max_sweeps, int32_t* sweeps_out, double* residual_out,
                           double* copy_to);

// Option "autotune": time every execution path this engine is eligible for ONCE, on the evidence in force, and keep the fastest
// for all later runs (the built-in choice between them rests on thresholds measured on a handful of networks on one pool of
// machines).  A trial is one run capped at 6 sweeps, evidence staged, host wall clock, best of two after one warm-up.  The
// choice is expressed through the engine's own options ("multisweep", "small", "mid", "dag"), so bn_set_option can still
// override it.  Paths whose >= 3-parent arithmetic differs in the last bits (bn_mi355x.h) may be exchanged by this.
static int autotune_paths(bn_engine* e, double eps) {
    struct Cand { int path, multisweep, small, mid, dag; bool ok; };
    const Cand cands[] = {
        {0, 0, 0, 0, 0, true},                               // one launch per sweep
        {2, 2, 0, 0, 0, e->resident_ok},                     // resident tiles
        {3, 1, 2, 0, 0, e->small_ok},                        // one workgroup, state in LDS
        {4, 1, 0, 2, 0, e->mid_ok},                          // the same items over several workgroups
        {5, 1, 0, 0, 2, e->dag_ok},                          // register-resident child tiles + parent items
    };
    const int keep[4] = {e->multisweep, e->small_mode, e->mid_mode, e->dag_mode};
    const bool keep_timing = e->timing;
    e->timing = false;
    double best = 1e300;
    int best_i = -1;
    for (int i = 0; i < 5; ++i) {
        if (!cands[i].ok) continue;
        e->multisweep = cands[i].multisweep; e->small_mode = cands[i].small; e->mid_mode = cands[i].mid; e->dag_mode = cands[i].dag;
        double t_best = 1e300;
        bool took = true;
        for (int rep = 0; rep < 3 && took; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = run_device_impl(e, eps, 6, nullptr, nullptr, nullptr);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rc != BN_OK) { e->multisweep = keep[0]; e->small_mode = keep[1]; e->mid_mode = keep[2]; e->dag_mode = keep[3]; e->timing = keep_timing; return rc; }
            took = e->last_path == cands[i].path;        // (a path in its pause after an abort, or refused by a policy: not a candidate now)
            if (rep > 0 && took) t_best = std::min(t_best, dt);
        }
        if (took && t_best < best) { best = t_best; best_i = i; }
    }
    e->timing = keep_timing;
    if (best_i < 0) { e->multisweep = keep[0]; e->small_mode = keep[1]; e->mid_mode = keep[2]; e->dag_mode = keep[3]; return BN_OK; }
    e->multisweep = cands[best_i].multisweep; e->small_mode = cands[best_i].small; e->mid_mode = cands[best_i].mid; e->dag_mode = cands[best_i].dag;
    e->autotuned_path = cands[best_i].path;
    if (std::getenv("BN_DEBUG")) std::fprintf(stderr, "[bn_mi355x] autotune: path %d (%.1f us per 6-sweep run)\n", e->autotuned_path, best * 1e6);
    return BN_OK;
}


// resident tiles pay on one block (no grid barrier at all) and on large networks (the CPT traffic saved outweighs the barrier); with 8
// waves per block the crossover was measured at ~600 tiles (160x160 grid, 402 tiles: 8.2 vs 8.9 us per sweep; 200x200, 627: 9.5 vs
// 9.2); at 4 waves per block (networks up to ~900 tiles: every wave has a SIMD of its own) it is faster than the launches from the
// smallest multi-block network on (32x32 grid 7.2 vs 7.4-7.8, 128x128 7.7 vs 8.0, 200x200 8.7 vs 9.5).  Shards: the in-kernel exchange
// wherever every rank's tiles qualify and the peers are mapped ("multisweep" 0 = per-sweep launches + one RCCL all-gather per sweep).
static bool resident_wanted(const bn_engine* e) {
    constexpr int64_t kResidentMinTiles = 600;
    if (e->plan.nranks > 1) return e->shard_flow_ok && e->multisweep != 0;
    const bool pays = e->grid_resident == 1 || e->resident_waves < kResidentWaves || int64_t(e->plan.tiles.size()) >= kResidentMinTiles;
    return e->resident_ok && (e->multisweep == 2 || (e->multisweep == 1 && pays));
}
static int resident_gave_up(bn_engine* e) {
    ++e->resident_aborts;
    if (e->plan.nranks > 1) {
        // Sharded engines: NO unilateral fall-back inside the library.  A peer whose service block had already published the
        // final verdict may have returned BN_OK: it would never enter the RCCL all-gather this rank would now wait in, and
        // peers may still be storing into this rank's exchange region.  The caller's control plane decides for ALL ranks
        // (multigpu.run_collective: all-reduce of the outcome, then "multisweep" 0 everywhere, or a collective retry);
        // nothing of the engine's state has been touched.
        const std::string why = g_err;
        return fail(BN_ERR_STATE, "the in-kernel exchange gave up a bounded wait on this rank (" + why + "): every rank must switch together -- "
                                  "set \"multisweep\" 0 on ALL ranks (RCCL exchange) or retry collectively");
    }
    // this run and the next few go down the per-sweep launches (8, 16, ... 1 024 runs), then the path is tried again
    e->resident_cooldown = e->resident_backoff;
    e->resident_backoff = std::min(e->resident_backoff * 2, 1024);
    report_abort_once(e, "the resident-tile kernel (bn_resident.hip)", e->resident_cooldown);
    return BN_OK;
}
void bn_eng::resident_ran_ok(bn_engine* e) { e->resident_backoff = 8; }

// The one-workgroup path is taken wherever the network fits, except where the resident-tile kernel runs the network in ONE block and
// was measured faster (scripts/experiments/small_vs_resident.py, us per sweep small / resident): chains and trees (one parent per
// node) beyond ~128 nodes or one round of entry items (200-node chain, k = 4: 5.2 / 2.8; 100 nodes: 2.9 / 2.6), and networks that
// need two rounds of accumulator or product items (16 x 16 grid, k = 2: 4.3 / 3.5).  With two parents per node the tile kernel's
// 64-entry contraction costs more than the items (8 x 8 grid, k = 4: 4.2 / 5.1; 40-node DAG: 2.6 / 6.4).
static bool small_wanted(const bn_engine* e) {
    if (!e->small_ok || e->multisweep == 0 || e->small_mode == 0) return false;
    if (e->small_mode == 2) return true;
    return !(e->resident_ok && e->grid_resident == 1) ||
           (e->small.rb == 1 && e->small.rc == 1 && (e->small.mmax >= 2 || (e->small.re == 1 && e->small.n <= 128)));
}
int bn_eng::small_gave_up(bn_engine*) { return BN_OK; }   // (one workgroup: it waits for nobody)

// the register-resident DAG path AHEAD of the one-workgroup path: forced ("dag" 2), or a small network of three or more rounds of
// entry items (dag_applies has the measurements)
static bool dag_first_wanted(const bn_engine* e) { return (e->dag_mode == 2 || (e->small_ok && e->small_mode != 2)) && dag_applies(e); }
static bool dag_later_wanted(const bn_engine* e) { return !dag_first_wanted(e) && dag_applies(e); }
static int dag_gave_up(bn_engine* e) {
    ++e->dag_aborts;
    e->dag_cooldown = 64;   // something else holds CUs: the other paths for a while
    report_abort_once(e, "the register-resident DAG kernel (bn_dag.hip)", 64);
    return BN_OK;
}
static int mid_gave_up(bn_engine* e) {
    ++e->mid_aborts;
    e->mid_cooldown = 64;
    report_abort_once(e, "the several-workgroup item kernel (bn_mid.hip)", 64);
    return BN_OK;
}

static const PathDriver kOneLaunchPaths[] = {
    {5, dag_first_wanted, run_dag, dag_gave_up, nullptr, &bn_engine::dag_cooldown, false},
    {3, small_wanted, run_small, small_gave_up, nullptr, &bn_engine::small_cooldown, false},
    {5, dag_later_wanted, run_dag, dag_gave_up, nullptr, &bn_engine::dag_cooldown, false},   // (its place by default: behind the one-workgroup path)
    {4, mid_applies, run_mid, mid_gave_up, nullptr, &bn_engine::mid_cooldown, false},
    {2, resident_wanted, run_resident, resident_gave_up, resident_ran_ok, &bn_engine::resident_cooldown, true},
};

static int run_device_impl(bn_engine* e, double eps, int32_t max_sweeps, int32_t* sweeps_out, double* residual_out,
                           double* copy_to) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (e->autotune_pending && !e->host_only && e->plan.nranks == 1) {
        e->autotune_pending = false;
        double* const keep_override = e->beliefs_override;
        e->beliefs_override = nullptr;          // (trial runs write into the engine's own buffer)
        const int rc = autotune_paths(e, eps);
        e->beliefs_override = keep_override;
        if (rc != BN_OK) return rc;
    }
    e->beliefs_on_host_only = false;  // (bn_bp_run_view sets it again when its kernels wrote to the host buffer)
    if (e->host_only) return fail(BN_ERR_STATE, "engine was created with BN_DEVICE_HOST_ONLY: no GPU, no compute");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: bn_reload_cpt failed while uploading (destroy it and create a new one)");
    if (max_sweeps < 0) return fail(BN_ERR_ARG, "max_sweeps < 0");
    if (e->plan.nranks > 1 && !e->comm && !(e->shard_flow_ok && e->multisweep != 0))
        return fail(BN_ERR_COMM, "sharded engine: call bn_comm_init (RCCL exchange) or bn_peer_import (in-kernel exchange) before running");
    const auto t_begin = std::chrono::steady_clock::now();
    ON_DEVICE(e);
    hipStream_t s = e->stream;
    int rc;
    if (e->plan.nranks > 1) ++e->shard_run_seq;
    // The one-launch paths, in the order of kOneLaunchPaths: the first one that wants the network (eligible, and chosen by the
    // options / the measured defaults) and is not paused runs the query; one that gives up a bounded wait pauses itself and
    // hands the query to the next; what none of them takes runs with one launch per sweep (below).
    bool evidence_flushed = false;
    for (const PathDriver& d : kOneLaunchPaths) {
        if (!d.wanted(e)) continue;
        if (d.reads_tile_evidence && !evidence_flushed) {   // the tile kernels read the evidence from their own buffers
            if ((rc = flush_evidence(e))) return rc;
            evidence_flushed = true;
        }
        int32_t& cooldown = e->*(d.cooldown);
        if (cooldown > 0) { --cooldown; continue; }   // paused after a launch that gave up
        rc = d.run(e, eps, max_sweeps, copy_to);
        if (rc == BN_OK) {
            if (d.ran_ok) d.ran_ok(e);
            e->stats.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
            if (sweeps_out) *sweeps_out = e->last_ctl.n_sweeps;
            if (residual_out) *residual_out = e->last_ctl.last_res;
            return BN_OK;
        }
        if (rc != BN_ERR_STATE) return rc;
        if ((rc = d.gave_up(e)) != BN_OK) return rc;   // counters, pause, one line on stderr (a shard: an error, see resident_gave_up)
    }
    if (!evidence_flushed && (rc = flush_evidence(e))) return rc;
    e->last_path = 0;
    if (e->plan.nranks > 1 && !e->comm)
        return fail(BN_ERR_COMM, "the in-kernel exchange gave up and no RCCL communicator is set up to fall back on (bn_comm_init)");
    if ((rc = step_begin(e))) return rc;
    int32_t launched = 0, batches = 0;
    // every rank takes the same decisions: they all see the same sweep counts
    int32_t batch = e->predicted_sweeps > 0 ? e->predicted_sweeps : 8;
    for (;;) {
        if (max_sweeps > 0) batch = std::min(batch, max_sweeps - launched);
        if (e->timing) {
            if ((rc = ensure_events(e, 2 * size_t(batches + 1)))) return rc;
            HIPCHK(hipEventRecord(e->events[2 * batches], s));
        }
        const bool overlapped = e->plan.nranks > 1 && e->overlap && e->comm_stream;
        for (int32_t i = 0; i < batch; ++i) {
            if (overlapped) {
                if ((rc = step_sweep_overlapped(e, launched + i, eps, launched + i > 0))) return rc;
            } else {
                if ((rc = step_sweep(e, launched + i, eps))) return rc;
                if ((rc = step_exchange(e, launched + i, s))) return rc;
            }
        }
        if (overlapped && batch > 0) HIPCHK(hipStreamWaitEvent(s, e->ev_gathered, 0));  // the finish kernel reads every rank's slots
        launched += batch;
        if (e->timing) HIPCHK(hipEventRecord(e->events[2 * batches + 1], s));
        ++batches;
        if ((rc = step_finish(e, launched, max_sweeps > 0 && launched >= max_sweeps, eps))) return rc;
        if (copy_to)  // behind the finish kernel; repeated if the predicted sweep count turns out too low
            HIPCHK(hipMemcpyAsync(copy_to, e->d_beliefs, sizeof(double) * e->plan.node_off[e->plan.n], hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        e->ev_upload_pending = false;
        if (e->h_ctl->run_id != e->run_id) return fail(BN_ERR_STATE, "finish kernel did not report (stale control block)");
        if (e->h_ctl->done != 0) break;
        batch = 8;
    }
    note_run_result(e);
    float ms = 0.f;
    for (int32_t i = 0; e->timing && i < batches; ++i) {
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, e->events[2 * i], e->events[2 * i + 1]));
        ms += t;
    }
    e->stats.sweep_launches = launched;
    e->stats.sweep_kernel_ms = ms;
    e->stats.total_ms =
        std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (sweeps_out) *sweeps_out = e->last_ctl.n_sweeps;
    if (residual_out) *residual_out = e->last_ctl.last_res;
    return BN_OK;
}

extern "C" int bn_bp_run_device(bn_engine* e, double eps, int32_t max_sweeps, int32_t* sweeps_out,
                                double* residual_out) {
    return run_device_impl(e, eps, max_sweeps, sweeps_out, residual_out, nullptr);
}

// Engine options: "timing" = 1/0 (HIP events around the sweep launches; off: sweep_kernel_ms reads 0);
// "multisweep" = 1/0 (networks small enough run all their sweeps in ONE launch; 0 forces one launch per sweep).
extern "C" int bn_set_option(bn_engine* e, const char* name, int32_t value) {
    if (!e || !name) return fail(BN_ERR_ARG, "null argument");
    if (std::strcmp(name, "timing") == 0) { e->timing = value != 0; return BN_OK; }
    if (std::strcmp(name, "overlap") == 0) { e->overlap = value != 0; return BN_OK; }
    if (std::strcmp(name, "beliefs_direct") == 0) { e->beliefs_direct = value != 0; return BN_OK; }
    if (std::strcmp(name, "flow") == 0) { e->flow = value != 0; return BN_OK; }
    if (std::strcmp(name, "direct") == 0) { e->resident_direct = value != 0; return BN_OK; }
    if (std::strcmp(name, "mid") == 0) { e->mid_mode = value < 0 ? 0 : (value > 2 ? 2 : value); return BN_OK; }
    if (std::strcmp(name, "dagflow") == 0) { e->dag_flow = value != 0; return BN_OK; }
    if (std::strcmp(name, "dag") == 0) { e->dag_mode = value < 0 ? 0 : (value > 2 ? 2 : value); return BN_OK; }
    if (std::strcmp(name, "autotune") == 0) { e->autotune_pending = value != 0; if (value == 0) e->autotuned_path = -1; return BN_OK; }
    if (std::strcmp(name, "small") == 0) { e->small_mode = value < 0 ? 0 : (value > 2 ? 2 : value); return BN_OK; }
    if (std::strcmp(name, "poll_sleep") == 0) { e->poll_sleep = std::max(0, std::min(value, 64)); return BN_OK; }
    if (std::strcmp(name, "multisweep") == 0) { e->multisweep = value < 0 ? 0 : (value > 2 ? 2 : value); return BN_OK; }
    return fail(BN_ERR_ARG, std::string("unknown option ") + name);
}
// Introspection for tests and tools: a named integer property of the engine / its last run.
extern "C" int64_t bn_get_info(bn_engine* e, const char* name) {
    if (!e || !name) return fail(BN_ERR_ARG, "null argument");
    if (std::strcmp(name, "resident_eligible") == 0) return e->resident_ok ? 1 : 0;
    if (std::strcmp(name, "flow_eligible") == 0) return e->flow_ok ? 1 : 0;
    if (std::strcmp(name, "last_flow") == 0) return e->last_path == 2 ? e->last_flow : 0;
    if (std::strcmp(name, "nbr_max") == 0) return e->plan.nbr_max;
    if (std::strcmp(name, "nbr_chunks") == 0) return e->plan.nbr.empty() ? 0 : e->plan.nbr_chunks;
    if (std::strcmp(name, "shard_flow") == 0) return e->shard_flow_ok ? 1 : 0;
    if (std::strncmp(name, "create_us_", 10) == 0) {   // construction split: host plans (tile / one-workgroup / several-workgroup / DAG), device side
        static const char* const kPhase[] = {"plan", "small", "mid", "dag", "device"};
        for (int i = 0; i < 5; ++i)
            if (std::strcmp(name + 10, kPhase[i]) == 0) return e->create_us[i];
        return fail(BN_ERR_ARG, std::string("unknown info ") + name);
    }
    if (std::strcmp(name, "rccl_ranks") == 0) {   // what the communicator itself reports (0: none initialised)
        int n = 0;
        if (e->comm && g_rccl.CommCount && g_rccl.CommCount(e->comm, &n) != ncclSuccess) n = -1;
        return n;
    }
    if (std::strcmp(name, "n_boundary_nodes") == 0) return int64_t(e->plan.boundary_node.size());
    if (std::strcmp(name, "resident_blocks") == 0) return e->grid_resident;
    if (std::strcmp(name, "resident_waves") == 0) return e->resident_waves;
    if (std::strcmp(name, "resident_aborts") == 0) return e->resident_aborts;
    if (std::strcmp(name, "mid_eligible") == 0) return e->mid.ok ? 1 : 0;
    if (std::strcmp(name, "mid_parts") == 0) return e->mid.ok ? int64_t(e->mid.parts.size()) : 0;
    if (std::strcmp(name, "mid_aborts") == 0) return e->mid_aborts;
    if (std::strcmp(name, "autotuned") == 0) return e->autotuned_path >= 0 ? 1 : 0;
    if (std::strcmp(name, "autotuned_path") == 0) return e->autotuned_path >= 0 ? e->autotuned_path : 0;   // (valid when "autotuned" is 1)
    if (std::strcmp(name, "dag_eligible") == 0) return e->dag.ok ? 1 : 0;
    if (std::strcmp(name, "dag_blocks") == 0) return e->dag.ok ? e->dag.blocks : 0;
    if (std::strcmp(name, "dag_tiles") == 0) return e->dag.ok ? int64_t(e->dag.tiles.size()) : 0;
    if (std::strcmp(name, "dag_stream") == 0) return e->dag.ok && e->dag.stream ? 1 : 0;
    if (std::strcmp(name, "lw_small") == 0) return e->lw.ready && e->lw.small ? 1 : 0;   // (known after the first sampler call)
    if (std::strcmp(name, "dag_aborts") == 0) return e->dag_aborts;
    if (std::strcmp(name, "batch_dense_refused") == 0) return e->dense_refused ? 1 : 0;   // (known after the first batch of >= 2 sets)
    if (std::strcmp(name, "batch_on_dense") == 0) return e->batch_on_dense ? 1 : 0;
    if (std::strcmp(name, "dag_flow_eligible") == 0) return e->dag_flow_ok ? 1 : 0;     // (known once the path has been set up: ensure_dag)
    if (std::strcmp(name, "dag_flow_max_nbr") == 0) return e->dag_flow_max_nbr;
    if (std::strcmp(name, "last_dag_flow") == 0) return e->last_path == 5 ? e->last_dag_flow : 0;
    if (std::strcmp(name, "small_eligible") == 0) return e->small.ok ? 1 : 0;
    if (std::strcmp(name, "small_waves") == 0) return e->small.ok ? e->small.waves : 0;
    if (std::strcmp(name, "small_lds_bytes") == 0) return e->small.ok ? int64_t(e->small.lds_bytes) : 0;
    return fail(BN_ERR_ARG, std::string("unknown info ") + name);
}
// 0 per-sweep launches, 2 resident tiles (bn_resident.hip), 3 one workgroup with the state in LDS (bn_small.hip), 4 the same items over several workgroups (bn_mid.hip)
extern "C" int bn_bp_last_path(bn_engine* e) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    return e->last_path;
}

// ---- single steps (diagnostics / tests) -------------------------------------------------------
extern "C" int bn_bp_step_begin(bn_engine* e) {
    if (!e || e->host_only) return fail(BN_ERR_STATE, "no device engine");
    ON_DEVICE(e);
    return step_begin(e);
}
extern "C" int bn_bp_step_sweep(bn_engine* e, int32_t sweep, double eps) {
    if (!e || e->host_only) return fail(BN_ERR_STATE, "no device engine");
    ON_DEVICE(e);
    return step_sweep(e, sweep, eps);
}
// part 1: the interior tiles only; part 2: the tiles that touch a cut edge + the residual bookkeeping
// (together one sweep; the overlapped run launches part 1 before the previous sweep's exchange has landed)
extern "C" int bn_bp_step_sweep_part(bn_engine* e, int32_t sweep, double eps, int32_t part) {
    if (!e || e->host_only) return fail(BN_ERR_STATE, "no device engine");
    if (part < 0 || part > 2) return fail(BN_ERR_ARG, "part must be 0, 1 or 2");
    ON_DEVICE(e);
    return step_sweep(e, sweep, eps, part);
}
extern "C" int bn_bp_step_finish(bn_engine* e, int32_t launched, int32_t final_batch, double eps, int32_t* done_out,
                                 int32_t* sweeps_out, double* residual_out) {
    if (!e || e->host_only) return fail(BN_ERR_STATE, "no device engine");
    ON_DEVICE(e);
    int rc = step_finish(e, launched, final_batch != 0, eps);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->h_ctl->run_id != e->run_id) return fail(BN_ERR_STATE, "finish kernel did not report (stale control block)");
    if (done_out) *done_out = e->h_ctl->done;
    if (sweeps_out) *sweeps_out = e->h_ctl->n_sweeps;
    if (residual_out) *residual_out = e->h_ctl->last_res;
    if (e->h_ctl->done != 0) note_run_result(e);
    return BN_OK;
}
// the marginals of the last run in device memory (a bn_bp_run_view that wrote them straight to the host buffer: uploaded first)
static int beliefs_to_device(bn_engine* e) {
    if (!e->beliefs_on_host_only) return BN_OK;
    ON_DEVICE(e);
    HIPCHK(hipMemcpyAsync(e->d_beliefs, e->h_beliefs, sizeof(double) * e->plan.node_off[e->plan.n], hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->beliefs_on_host_only = false;
    return BN_OK;
}
extern "C" const double* bn_bp_beliefs_device(bn_engine* e) {
    if (!e || e->host_only) return nullptr;
    if (beliefs_to_device(e) != BN_OK) return nullptr;
    return e->d_beliefs;
}

extern "C" int bn_bp_copy_beliefs(bn_engine* e, double* beliefs_out) {
    if (!e || !beliefs_out) return fail(BN_ERR_ARG, "null argument");
    if (e->host_only || !e->have_run) return fail(BN_ERR_STATE, "no belief propagation run to copy from");
    if (e->beliefs_on_host_only) {  // the last run's marginals are in the engine's page-locked buffer
        if (beliefs_out != e->h_beliefs) std::memcpy(beliefs_out, e->h_beliefs, sizeof(double) * e->plan.node_off[e->plan.n]);
        return BN_OK;
    }
    ON_DEVICE(e);
    HIPCHK(hipMemcpyAsync(beliefs_out, e->d_beliefs, sizeof(double) * e->plan.node_off[e->plan.n],
                          hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return BN_OK;
}

// Evidence in, beliefs out, ONE stream synchronisation: the evidence upload, the evidence kernel, the run and
// the copy of the beliefs are queued back to back on the engine's stream and waited for once.
extern "C" int bn_bp_run(bn_engine* e, int32_t ne, const int32_t* ev_node, const int32_t* ev_off,
                         const double* ev_val, double eps, int32_t max_sweeps, double* beliefs_out,
                         int32_t* sweeps_out, double* residual_out) {
    if (!beliefs_out) return fail(BN_ERR_ARG, "null beliefs_out");
    int rc = set_evidence_impl(e, ne, ev_node, ev_off, ev_val, false);
    if (rc) return rc;
    return run_device_impl(e, eps, max_sweeps, sweeps_out, residual_out, beliefs_out);
}

// The same with the beliefs left in a pinned host buffer the engine owns (valid until the next run on this
// engine): the copy behind the run is one DMA into page-locked memory, and a caller that unpacks the flat
// array anyway (the C++ functor builds its map of 1 x k matrices from it) never needs a second copy.
extern "C" int bn_bp_run_view(bn_engine* e, int32_t ne, const int32_t* ev_node, const int32_t* ev_off,
                              const double* ev_val, double eps, int32_t max_sweeps, const double** beliefs_view,
                              int32_t* sweeps_out, double* residual_out) {
    if (!beliefs_view) return fail(BN_ERR_ARG, "null beliefs_view");
    *beliefs_view = nullptr;
    int rc = set_evidence_impl(e, ne, ev_node, ev_off, ev_val, false);
    if (rc) return rc;
    if (!e->h_beliefs) {
        ON_DEVICE(e);
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->h_beliefs), std::max<size_t>(e->plan.node_off[e->plan.n], 1) * sizeof(double),
                             hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_beliefs_dev), e->h_beliefs, 0));
    }
    if (e->beliefs_direct && e->plan.nranks == 1 && e->plan.node_off[e->plan.n] * 8 <= (int64_t(16) << 20)) {
        e->beliefs_override = e->h_beliefs_dev;
        rc = run_device_impl(e, eps, max_sweeps, sweeps_out, residual_out, nullptr);
        e->beliefs_override = nullptr;
        e->beliefs_on_host_only = rc == BN_OK;
    } else {
        rc = run_device_impl(e, eps, max_sweeps, sweeps_out, residual_out, e->h_beliefs);
    }
    if (rc) return rc;
    *beliefs_view = e->h_beliefs;
    return BN_OK;
}

extern "C" int bn_bp_residual_history(bn_engine* e, double* out, int32_t cap) {
    if (!e || !out || cap < 0) return fail(BN_ERR_ARG, "bad argument");
    if (e->host_only || !e->have_run) return fail(BN_ERR_STATE, "no belief propagation run yet");
    int32_t cnt = std::min({cap, e->last_ctl.n_sweeps, e->res_cap});
    ON_DEVICE(e);
    if (cnt > 0) HIPCHK(hipMemcpy(out, e->d_res_hist, sizeof(double) * cnt, hipMemcpyDeviceToHost));
    return cnt;
}

extern "C" int bn_bp_messages(bn_engine* e, double* pi_msg_out, double* lambda_msg_out) {
    if (!e || !pi_msg_out || !lambda_msg_out) return fail(BN_ERR_ARG, "null argument");
    if (e->host_only || !e->have_run) return fail(BN_ERR_STATE, "no belief propagation run yet");
    ON_DEVICE(e);
    if (e->last_path == 5) {  // bn_dag.hip: CSR edge order, four doubles per edge, two buffers: the run stopped in buffer n_sweeps & 1
        const int64_t E = e->dag.E, n = e->dag.n;
        const int par = e->last_ctl.n_sweeps & 1;
        // the records (tile-major in the state, bn_dag.hpp) back in CSR edge order; arities below 4: of a padded record edge e's first
        // k(parent of e) entries exist
        std::vector<double> pm(size_t(E) * 4), lm(size_t(E) * 4);
        HIPCHK(hipMemcpy(pm.data(), e->d_g_state + 2 * dag_off_pim(E, n, par, 0), sizeof(double) * 4 * size_t(E), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(lm.data(), e->d_g_state + 2 * dag_off_lam(E, n, par, 0), sizeof(double) * 4 * size_t(E), hipMemcpyDeviceToHost));
        size_t at = 0;
        for (int64_t ed = 0; ed < E; ++ed) {
            const int kp = e->dag.uniform4 ? 4 : e->plan.k[e->plan.in_idx[ed]];
            const size_t rec = size_t(e->dag_tables.eperm[size_t(ed)]);
            for (int i = 0; i < kp; ++i, ++at) { pi_msg_out[at] = pm[rec * 4 + i]; lambda_msg_out[at] = lm[rec * 4 + i]; }
        }
        return BN_OK;
    }
    if (e->last_path == 4) {  // bn_mid.hip keeps them in CSR edge order, two buffers: the run stopped in buffer n_sweeps & 1
        const size_t M = size_t(e->mid.parts[0].M), par = size_t(e->last_ctl.n_sweeps & 1);
        HIPCHK(hipMemcpy(pi_msg_out, e->d_m_state + par * M, sizeof(double) * M, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(lambda_msg_out, e->d_m_state + 2 * M + par * M, sizeof(double) * M, hipMemcpyDeviceToHost));
        return BN_OK;
    }
    if (e->last_path == 3) {  // bn_small.hip leaves the messages in CSR edge order
        const size_t bytes = sizeof(double) * size_t(e->small.M);
        HIPCHK(hipMemcpy(pi_msg_out, e->d_s_state, bytes, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(lambda_msg_out, e->d_s_state + e->small.M, bytes, hipMemcpyDeviceToHost));
        return BN_OK;
    }
    std::vector<double> rec(std::max<int64_t>(e->plan.rec_total_doubles, 1));
    HIPCHK(hipMemcpy(rec.data(), e->d_rec[e->last_ctl.n_sweeps & 1], sizeof(double) * e->plan.rec_total_doubles,
                     hipMemcpyDeviceToHost));
    unstripe_messages(e->plan, rec, pi_msg_out, lambda_msg_out);
    return BN_OK;
}

extern "C" int bn_bp_last_stats(bn_engine* e, bn_bp_stats* out) {
    if (!e || !out) return fail(BN_ERR_ARG, "null argument");
    *out = e->stats;
    out->resident_aborts = e->resident_aborts;
    return BN_OK;
}


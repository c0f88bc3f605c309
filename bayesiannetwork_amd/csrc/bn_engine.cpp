// bn_engine.cpp -- C ABI (include/bn_mi355x.h) over the HIP kernels: device memory, the run
// loop of belief propagation, diagnostics.  No CPU compute path exists here: every result
// comes from the kernels in bn_kernels.hip / bn_lw_kernels.hip.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <rccl/rccl.h>  // types only: the library is loaded lazily with dlopen (no link dependency)

#include "bn_device.hpp"
#include "bn_fit.hpp"
#include "bn_lw.hpp"
#include "bn_small.hpp"
#include "bn_dag.hpp"

using namespace bnmi;

static thread_local std::string g_err;

// Entry points run on the engine's device and leave the calling thread's current device as they
// found it (a caller may drive another GPU from the same thread).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t enter(int device) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev == device) return hipSuccess;
        e = hipSetDevice(device);
        switched = e == hipSuccess;
        return e;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define ON_DEVICE(e)            \
    DeviceGuard guard_;         \
    HIPCHK(guard_.enter((e)->device))

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(BN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

// RCCL entry points, resolved at the first bn_comm_* call.  In a process that already loaded
// librccl.so.1 (e.g. through torch) the same copy is reused.
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static int load_rccl() {
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
    if (!a.AllReduce || !a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString)
        return fail(BN_ERR_COMM, "librccl lacks a required symbol");
    g_rccl = a;
    return BN_OK;
}

struct bn_engine {
    Plan plan;
    bool host_only = true;
    bool poisoned = false;          // a bn_reload_cpt upload failed half-way: device images of mixed age, every compute call is refused
    int device = -1;
    hipStream_t stream = nullptr;
    // device images
    TileDesc* d_tiles = nullptr;
    ClassDesc* d_classes = nullptr;
    FlatEntry* d_flat_tab = nullptr;
    double* d_cpt = nullptr;
    double* d_rec[2] = {nullptr, nullptr};
    double* d_node[2] = {nullptr, nullptr};
    MsgRef* d_out = nullptr;
    MsgRef* d_inrefs = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t comm_stream = nullptr;   // sharded runs: the all-gathers run here, beside the interior tiles' launch
    hipEvent_t ev_swept = nullptr;       // main stream: every tile of the current sweep has been launched
    hipEvent_t ev_gathered = nullptr;    // comm stream: the current sweep's all-gather
    bool overlap = true;                 // BN_OVERLAP=0 / bn_set_option("overlap", 0): kernel and collective back to back
    uint8_t* d_frozen = nullptr;
    uint8_t frozen_mark = 1;        // mark value of the evidence set in force (1..255; wrapping clears the array)
    char* h_ev_dev = nullptr;       // h_ev as the device sees it (mapped page-locked memory: the evidence kernel reads it in place)
    int32_t* d_slot_node = nullptr;
    int64_t* d_slot_boff = nullptr;
    int32_t* d_node_tile = nullptr;
    int32_t* d_node_nl = nullptr;
    double* d_res_hist = nullptr;
    Ctl* d_ctl = nullptr;
    double* d_beliefs = nullptr;
    // evidence staging: one device block + one pinned host block, sub-pointers into d_ev
    char* d_ev = nullptr;
    char* h_ev = nullptr;
    size_t ev_bytes_cap = 0;
    int32_t ev_ne = 0;
    int32_t ev_nval = 0;            // values of the evidence in force (sum of the observed nodes' arities)
    int32_t* d_ev_node = nullptr;
    int32_t* d_ev_off = nullptr;
    double* d_ev_val = nullptr;
    bool ev_applied_dirty = false;  // an evidence launch failed: marks unknown, clear them at the next set
    bool rows_clean = true;         // residual slots are zero (left so by the last finish kernel / the reset kernel)
    uint32_t run_id = 0;            // id of the current / last run (bn_device.hpp Ctl)
    bool nontemporal = false;
    bool timing = false;            // HIP events around each batch of sweeps (bn_bp_stats.sweep_kernel_ms); opt-in:
                                    // an event record between two launches opens a ~6 us bubble in the queue
    bool resident_ok = false;       // every tile register-resident and co-resident: the whole run in one launch (bn_resident.hip)
    // A launch of the resident kernel that gives up a bounded wait (its blocks were not all co-resident: another
    // process or engine held CUs) sends this and the next `resident_cooldown` runs down the per-sweep launches;
    // after that the resident path is tried again, and a repeated abort doubles the pause (<= 1024 runs).
    int32_t resident_aborts = 0;    // launches that gave up, over the engine's life (bn_bp_stats.resident_aborts)
    int32_t resident_cooldown = 0;  // runs left before the resident path is tried again
    int32_t resident_backoff = 8;   // length of the next pause
    double* h_beliefs = nullptr;    // pinned: bn_bp_run_view hands this out, bn_bp_run stages nothing through it
    double* h_beliefs_dev = nullptr;  // ... as the device sees it
    double* beliefs_override = nullptr;  // where the kernels write the beliefs of the run in hand instead of d_beliefs
    bool beliefs_on_host_only = false;   // the last run wrote its marginals into h_beliefs, not d_beliefs (synced back on demand)
    int beliefs_direct = 1;         // option "beliefs_direct": bn_bp_run_view lets the kernels write the marginals straight into
                                    // the mapped host buffer (no copy command behind the run; 316x316 grid: 253 -> 235 us per query);
                                    // outputs above 16 MB go through the copy engine (larger PCIe payloads)
    std::vector<uint32_t> ev_seen;  // check_evidence: epoch stamp per node (no per-call allocation)
    uint32_t ev_epoch = 0;
    bool ev_upload_pending = false; // an evidence H2D from h_ev may still be in flight (no sync since)
    int resident_lean = 0;          // ... and every node has this arity (2, 3 or 4) and <= 2 children; else 0
    int grid_resident = 0;
    int resident_waves = kResidentWaves;  // tiles per block of the resident kernel (8, or 4 on networks small enough)
    int resident_poll_margin = 30;  // direct form: 10 ns ticks between the predicted arrival of the last block and a block's first poll (BN_RESIDENT_DELAY)
    int resident_direct = 1;        // option "direct" / BN_RESIDENT_DIRECT: the grid barrier without a service block (bn_resident.hip wait_verdict);
                                    // measured against the service block, us per sweep: 32 x 32 grid 6.98 -> 6.46, 128 x 128 7.25 -> 6.80, 316 x 316 11.46 -> 11.04
    ResidentSync* d_rsync = nullptr;
    bool rsync_dirty = true;        // the sync block must be zeroed before the next launch
    // dataflow form of the resident kernel (no grid barrier; single evidence set, more than one tile block)
    bool flow_ok = false;           // every tile has <= 64 neighbour tiles
    int poll_sleep = 2;             // option "poll_sleep" / BN_POLL_SLEEP: pause between two polls of a waiting tile (x 512 cycles)
    int flow = 0;                   // option "flow" / BN_RESIDENT_FLOW: 1 = dataflow form where eligible, 0 = grid barrier per sweep
                                    // (the default on one GPU: measured equal per sweep, and the lagging stop decision costs one
                                    // speculative iteration per run; sharded engines exchange through the dataflow form)
    FlowSync* d_flow = nullptr;
    bool flow_dirty = true;
    uint32_t flow_gen_base = 0;
    int32_t* d_nbr = nullptr;
    // sharded engines: halo exchange inside the resident kernel (bn_peer_export / bn_peer_import)
    bool shard_shapes_ok = false;   // this shard's tiles are what the resident kernel runs (uniform arity, <= 2 parents, <= 8 children)
    bool shard_flow_ok = false;     // ... on every rank, and the peers' buffers are mapped: the dataflow form exchanges in-kernel
    bool fine_grained = false;      // record buffers / sync block allocated fine-grained (peers store into them)
    uint32_t shard_run_seq = 0;     // bn_bp_run_device calls on this sharded engine: every rank counts alike -> same generations
    PeerTable* d_peers = nullptr;
    uint32_t* d_pub_mask = nullptr;
    std::vector<uint32_t> pub_mask; // host copy (introspection)
    std::vector<void*> ipc_opened;  // hipIpcOpenMemHandle results to close
    unsigned* h_abort = nullptr;    // pinned + mapped: set by a kernel that gives up a bounded wait
    unsigned* h_abort_dev = nullptr;
    uint32_t gen_base = 0;          // barrier generations used so far on d_rsync
    // several evidence sets per launch (bn_bp_*_batch): per-set records, node vectors, marks, beliefs, histories
    struct Batch {
        int32_t n_sets = 0, cap_sets = 0;
        double* d_rec[2] = {nullptr, nullptr};
        double* d_node[2] = {nullptr, nullptr};
        uint8_t* d_frozen = nullptr;
        double* d_beliefs = nullptr;
        double* d_res_hist = nullptr;
        ResidentSync* d_sync = nullptr;  // resident path: [min(cap_sets, kResidentMaxSets)]
        bool sync_dirty = true;
        uint32_t gen_base = 0;
        double* d_s_state = nullptr;  // one-workgroup path (bn_small.hip): [cap_sets][2 M + 2 N]
        // register-resident DAG path (bn_dag.hip), several sets per launch: [dag_sets] states, marks, barrier words (allocated at first use)
        double* d_g_state = nullptr;
        uint8_t* d_g_frz = nullptr;
        ResidentSync* d_g_sync = nullptr;
        int32_t dag_sets = 0;
        uint8_t dag_mark = 0;
        bool dag_sync_dirty = true;
        uint32_t dag_gen_base = 0;
        bool ev_deferred = false;     // the sets' evidence sits in d_ev only (read there by that kernel); d_ev_meta: per set {count, first node / offset / value}
        int32_t* d_ev_meta = nullptr;   // (inside the staging block)
        char* h_ev = nullptr;           // small networks: the staging block is page-locked host memory the kernels read in place
        char* ev_base = nullptr;        // the staging block as the device sees it: d_ev, or h_ev mapped
        size_t h_ev_cap = 0;
        double* h_beliefs = nullptr;    // small networks, bn_bp_run_batch: the kernel writes every set's marginals here (mapped) ...
        double* h_beliefs_dev = nullptr;
        size_t h_beliefs_cap = 0;
        bool direct_out = false;        // ... when this is set for the run at hand
        bool beliefs_on_host = false;   // the last run's marginals are in h_beliefs, not d_beliefs
        size_t ev_b_node = 0, ev_b_off = 0, ev_b_val = 0;  // where the three arrays start inside d_ev
        std::vector<int64_t> ev_node_at, ev_off_at, ev_val_at;
        Ctl* d_ctl = nullptr;       // per-sweep launches: one control block per set
        bool rows_clean = true;     // ... and every set's residual slots are zero
        int32_t predicted_sweeps = 0;
        Ctl* h_ctl = nullptr;       // pinned, [cap_sets]
        Ctl* h_ctl_dev = nullptr;
        char* d_ev = nullptr;       // staging of every set's evidence
        size_t ev_cap = 0;
        // host copy of the evidence (sets run one after another when the network is not resident-eligible)
        std::vector<int32_t> ne, ev_node, ev_off;
        std::vector<double> ev_val;
        std::vector<int32_t> sweeps;
        std::vector<double> residual;
        bool have_run = false;
    } batch;
    bool batch_on_dense = false;    // the current batch lives in `dense`
    bn_engine* dense = nullptr;     // a second engine with the dense layout: batches on a network whose own layout trades
                                    // wavefront count for one query's latency (Plan::latency_rules_applied) run there
    // small networks: the whole run in ONE workgroup with the state in LDS (bn_small.hip)
    SmallPlan small;
    bool small_ok = false;
    int small_mode = 1;             // option "small": 0 never, 1 where it was measured faster than the other paths, 2 wherever eligible
    SmallEntry* d_s_ent = nullptr;
    double* d_s_cpt = nullptr;
    uint32_t* d_s_term = nullptr;
    uint16_t* d_s_clist = nullptr;
    SmallSlot* d_s_bslot = nullptr;
    SmallSlot* d_s_cslot = nullptr;
    int32_t* d_s_nvidx = nullptr;
    int32_t* d_s_nvslot = nullptr;
    double* d_s_init = nullptr;
    double* d_s_state = nullptr;    // [2 M + 2 N] the state the last launch stopped in
    int32_t* d_s_nodeoff = nullptr;
    // networks beyond one workgroup's LDS, spread over up to 32 (bn_mid.hip): the same items, state in device memory
    MidPlan mid;
    bool mid_ok = false;
    int mid_mode = 1;               // option "mid": 0 never, 1 where eligible and the resident tiles do not cover the network, 2 wherever eligible
    int32_t small_cooldown = 0;     // (never set: the one-workgroup path waits for nobody; the path table wants a member)
    int32_t mid_cooldown = 0, mid_aborts = 0;   // runs left on the tile kernels after a grid wait gave up; how often that happened
    MidPart* d_m_parts = nullptr;
    SmallEntry* d_m_ent = nullptr;
    double* d_m_cpt = nullptr;
    uint32_t* d_m_term = nullptr;
    uint16_t* d_m_clist = nullptr;
    SmallSlot* d_m_bslot = nullptr;
    SmallSlot* d_m_cslot = nullptr;
    int32_t* d_m_nvidx = nullptr;
    int32_t* d_m_nvslot = nullptr;
    double* d_m_init = nullptr;
    int32_t* d_m_nodeoff = nullptr;
    int32_t* d_m_msgfirst = nullptr;
    double* d_m_state = nullptr;    // [4 M + 4 N]: pi[2][M], lam[2][M], npi[2][N], nlam[2][N]
    uint8_t* d_m_frz = nullptr;
    char* d_m_sync = nullptr;       // per state slot kMidSyncBytes: the barrier counter, the three residual words, the group counters
    int32_t mid_slots = 0;          // state slots allocated (1 for single queries; batches run several sets per launch)
    int32_t n_cus = 0;
    // k = 4 networks with up to 5 parents per node (BASELINE configs[1]): child tiles with the CPT in registers + parent items on
    // waves of their own, state in device memory, one launch per run (bn_dag.hip)
    DagPlan dag;
    bool dag_ok = false;
    int dag_mode = 1;               // option "dag": 0 never, 1 where eligible and no other one-launch path takes the network, 2 wherever eligible
    int32_t dag_cooldown = 0, dag_aborts = 0;   // runs left on the tile kernels after a grid wait gave up; how often that happened
    DagTile* d_g_tiles = nullptr;
    int32_t* d_g_slotptr = nullptr;
    DagChildLane* d_g_cnode = nullptr;
    DagParentLane* d_g_pitem = nullptr;
    int32_t* d_g_oedge = nullptr;
    double* d_g_cpt = nullptr;
    double* d_g_init = nullptr;
    int32_t* d_g_k = nullptr;       // networks with arities below 4 (DagPlan::uniform4 == false): arity, in-edge CSR and marginal offsets for the padded form
    int32_t* d_g_inptr = nullptr;
    int32_t* d_g_inidx = nullptr;
    int64_t* d_g_noff = nullptr;
    double* d_g_state = nullptr;    // pi-/lambda-messages (CSR edge order), pi(v), lambda(v): two buffers each (bn_dag.hpp dag_off_*)
    uint8_t* d_g_frz = nullptr;
    uint8_t dag_mark = 0;           // mark value of the evidence set applied to d_g_state / d_g_frz
    bool dag_ev_applied = false;    // ... and whether that is the set in force
    ResidentSync* d_g_sync = nullptr;
    bool dag_sync_dirty = true;
    uint32_t dag_gen_base = 0;
    bool ev_deferred = false;       // the evidence in force sits in the staging block only: the one-workgroup kernel reads it there
                                    // itself (no evidence launch in front of the run); the tile buffers get it -- marks, vectors --
                                    // when another path needs them (flush_evidence)
    bool autotune_pending = false;  // option "autotune": the next run first times every eligible path on the staged evidence and keeps the fastest
    int32_t autotuned_path = -1;    // ... the path it kept (bn_bp_last_path numbering), -1: never tuned
    bool abort_reported = false;    // the one stderr line about a one-launch path that gave up a bounded wait has been printed
    int multisweep = 1;             // resident one-launch path: 0 never, 1 where it was measured faster (one block, or
                                    // >= kResidentMinTiles tiles), 2 wherever eligible (tests, experiments)
    int32_t last_path = 0;          // 0 per-sweep launches, 2 one launch for the whole run (resident tiles), 3 one workgroup, state in LDS (bn_small.hip)
    int32_t last_flow = 0;          // ... in its dataflow form
    Ctl* h_ctl = nullptr;  // pinned
    Ctl* h_ctl_dev = nullptr;  // the same memory as the device sees it
    // run state
    int32_t res_cap = 1 << 16;
    int32_t predicted_sweeps = 0;
    bool have_run = false;
    Ctl last_ctl{};
    bn_bp_stats stats{};
    std::vector<hipEvent_t> events;  // (begin, end) per sweep batch
    int grid_tiles = 0;              // blocks for one-wave-per-tile kernels without remap
    LwState lw;
};

static void free_engine(bn_engine* e) {
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
                        e->d_g_tiles, e->d_g_slotptr, e->d_g_cnode, e->d_g_pitem, e->d_g_oedge, e->d_g_cpt, e->d_g_init, e->d_g_state, e->d_g_frz, e->d_g_sync, e->d_g_k, e->d_g_inptr, e->d_g_inidx, e->d_g_noff,
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
        if (e->stream) (void)hipStreamDestroy(e->stream);
    }
    delete e;
}

template <class T>
static int upload(T** dst, const std::vector<T>& src, hipStream_t s) {
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIPCHK(hipMalloc(reinterpret_cast<void**>(dst), bytes));
    if (!src.empty()) HIPCHK(hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return BN_OK;
}

template <class T>
static int dalloc(T** dst, size_t count) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(dst), std::max<size_t>(count, 1) * sizeof(T)));
    return BN_OK;
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
static int mid_reserve_slots(bn_engine* e, int32_t slots) {
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

static int create_impl(const bn_model_desc* desc, const ShardSpec& shard, bn_engine** out) {
    if (!desc || !out) return fail(BN_ERR_ARG, "null argument");
    if (std::getenv("BN_DEBUG")) signal(SIGSEGV, debug_segv_handler);
    *out = nullptr;
    bn_engine* e = new (std::nothrow) bn_engine();
    if (!e) return fail(BN_ERR_ALLOC, "out of host memory");
    if (const char* t = std::getenv("BN_TIMING")) e->timing = std::atoi(t) != 0;  // default off, see bn_engine::timing
    std::string err;
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
    if (p.nranks == 1 && !e->small.ok && !std::getenv("BN_NO_MID")) {  // ... spread over several workgroups (bn_mid.hip)
        try {
            build_mid_plan(p, e->mid);
        } catch (const std::bad_alloc&) {
            delete e;
            return fail(BN_ERR_ALLOC, "out of host memory while building the mid-size plan");
        }
    }
    constexpr int32_t kDagDefaultCap = 224;  // 0.9 x 256 CUs, a multiple of 8 (rebuilt below when the device has another count)
    if (p.nranks == 1 && !std::getenv("BN_NO_DAG")) {  // k = 4, <= 5 parents: register-resident child tiles + parent items (bn_dag.hpp)
        try {
            build_dag_plan(p, kDagDefaultCap, e->dag);
        } catch (const std::bad_alloc&) {   // the other paths can still run the network
            e->dag = DagPlan();
            e->dag.why = "out of host memory while building the plan";
        }
    }
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
    int rc = [&]() -> int {
        if (desc->device >= 0) {
            if (desc->device >= ndev) return fail(BN_ERR_ARG, "device ordinal out of range");
            e->device = desc->device;
        } else {
            HIPCHK(hipGetDevice(&e->device));
        }
        HIPCHK(guard.enter(e->device));
        HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
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
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->h_ctl), sizeof(Ctl), hipHostMallocMapped));
        std::memset(e->h_ctl, 0, sizeof(Ctl));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_ctl_dev), e->h_ctl, 0));
        {   // resident path (bn_resident.hip): one-lane tiles (<= 2 parents, <= 8 children per node), one wave per tile, every block co-resident (one 512-thread block of <= 256 VGPRs per CU)
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, e->device));
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
        if (e->dag.ok) {
            int32_t cap = int32_t((int64_t(e->n_cus) * 9 / 10) & ~int64_t(7));
            if (const char* c = std::getenv("BN_DAG_CAP")) cap = std::max(8, std::min(cap, std::atoi(c) & ~7));   // experiments: fewer blocks
            if (cap != kDagDefaultCap) {
                try {
                    build_dag_plan(p, cap, e->dag);
                } catch (const std::bad_alloc&) {
                    e->dag = DagPlan();
                    e->dag.why = "out of host memory while building the plan";
                }
            }
        }
        if (e->dag.ok) {
            const DagPlan& dp = e->dag;
            int r2;
            if ((r2 = upload(&e->d_g_tiles, dp.tiles, e->stream))) return r2;
            if ((r2 = upload(&e->d_g_slotptr, dp.slot_ptr, e->stream))) return r2;
            if ((r2 = upload(&e->d_g_cnode, dp.cnode, e->stream))) return r2;
            if ((r2 = upload(&e->d_g_pitem, dp.pitem, e->stream))) return r2;
            if ((r2 = upload(&e->d_g_oedge, dp.oedge, e->stream))) return r2;
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
            if (const char* dm = std::getenv("BN_DAG")) e->dag_mode = std::max(0, std::min(2, std::atoi(dm)));
            e->dag_ok = true;
        }
        if (const char* m = std::getenv("BN_MULTISWEEP")) e->multisweep = std::max(0, std::min(2, std::atoi(m)));
        HIPCHK(hipStreamSynchronize(e->stream));
        std::vector<double>().swap(e->plan.cpt_striped);  // the image now lives in HBM
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

static BpBuffers buffers_of(bn_engine* e) {
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
static int check_evidence(const Plan& p, int32_t ne, const int32_t* ev_node, const int32_t* ev_off,
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

static int ensure_events(bn_engine* e, size_t count) {
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

static bool dag_applies(const bn_engine* e);
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
static int resident_service_blocks(int tile_blocks) { return tile_blocks > 1 ? 1 : 0; }

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

static SmallArgs small_args_of(bn_engine* e, const BpBuffers& b, double eps, int32_t max_sweeps, int32_t begin, Ctl* host_ctl) {
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
static bool mid_applies(const bn_engine* e) {
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
static MidArgs mid_args_of(bn_engine* e, const BpBuffers& b0, const SetStrides& st, Ctl* h_ctl_dev, double eps, int32_t max_sweeps,
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
    a.sets = st; a.set_base = set_base; a.slot_base = slot_base;
    return a;
}
// launch + wait; BN_ERR_STATE: a grid wait gave up (the caller redoes the work on the tile kernels)
static int mid_launch(bn_engine* e, const MidArgs& a, int32_t n_sets, const double* copy_from, double* copy_to) {
    hipStream_t s = e->stream;
    *e->h_abort = 0;
    HIPCHK(hipMemsetAsync(e->d_m_sync + size_t(a.slot_base) * kMidSyncBytes, 0, size_t(n_sets) * kMidSyncBytes, s));
    if (int code = launch_bp_mid(a, e->mid.waves, e->mid.rounds, e->mid.lds_bytes, n_sets, s))
        return fail(BN_ERR_HIP, std::string("bp_mid launch failed: ") + hipGetErrorString(hipError_t(code)));
    if (copy_to) HIPCHK(hipMemcpyAsync(copy_to, copy_from, sizeof(double) * e->plan.node_off[e->plan.n], hipMemcpyDeviceToHost, s));
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
static bool dag_applies(const bn_engine* e) {
    if (!e->dag_ok || e->multisweep == 0 || e->dag_mode == 0) return false;
    if (e->dag_mode == 2) return true;
    // Networks the one-workgroup path (state in LDS) takes as well, us per query (profiles/r05_paths.json), that path / this one: one
    // round of entry items (ALARM-sized) 48.6 / 68.7, Pearl's four nodes 19.3 / 24.5 -- but 8 x 8 grid, k = 4 (four rounds) 86.0 / 64.4,
    // 60 nodes of mixed arity with <= 3 parents (three rounds) 72.3 / 64.6.  Chains and trees the resident tiles run in ONE block
    // stay there (200-node chain: 71.9 resident, 81.6 this path, 113.6 one workgroup).
    if (e->small_ok && e->small.re <= 2) return false;   // (two rounds: not measured; the one-workgroup path also keeps the reference's order for >= 3 parents)
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
    if (e->dag_ev_applied) return BN_OK;
    if (e->dag_mark == 255) {  // the mark values are used up: start over
        HIPCHK(hipMemsetAsync(e->d_g_frz, 0, size_t(e->dag.n), e->stream));
        e->dag_mark = 0;
    }
    ++e->dag_mark;
    DagEvidenceArgs ea{e->ev_ne, e->dag.n, e->dag.E, e->d_ev_node, e->d_ev_off, e->d_ev_val, e->d_g_state, e->d_g_frz, e->dag_mark, e->d_g_k};
    if (int code = launch_dag_evidence(ea, e->stream))
        return fail(BN_ERR_HIP, std::string("dag_evidence launch failed: ") + hipGetErrorString(hipError_t(code)));
    e->dag_ev_applied = true;
    e->ev_upload_pending = e->ev_ne > 0;
    return BN_OK;
}

// One launch runs the whole query (more only beyond kDagBudget iterations).  BN_ERR_STATE: a grid wait gave up.
static int run_dag(bn_engine* e, double eps, int32_t max_sweeps, double* copy_to) {
    hipStream_t s = e->stream;
    const DagPlan& dp = e->dag;
    if (int rc = flush_dag_evidence(e)) return rc;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    int32_t begin = 0, launches = 0;
    float ms = 0.f;
    double dev_ticks = 0.0;
    const BpBuffers b = buffers_of(e);
    if (!dp.uniform4) {   // arities below 4: the run's initial state stands in memory (zeros in the padding), bn_dag_plan.cpp
        DagInitArgs ia{dp.n, dp.E, e->d_g_inptr, e->d_g_inidx, e->d_g_k, e->d_g_init, e->d_g_state, e->d_g_frz, e->dag_mark};
        if (int code = launch_dag_init(ia, s))
            return fail(BN_ERR_HIP, std::string("dag_init launch failed: ") + hipGetErrorString(hipError_t(code)));
    }
    for (;;) {
        // polled words: generations count on from launch to launch; zeroed at creation, after an abort and before they would wrap
        if (e->dag_sync_dirty || e->dag_gen_base > (1u << 29)) {
            HIPCHK(hipMemsetAsync(e->d_g_sync, 0, sizeof(ResidentSync), s));
            e->dag_sync_dirty = false;
            e->dag_gen_base = 0;
        }
        *e->h_abort = 0;
        DagArgs a{};
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
            return fail(BN_ERR_STATE, "a block of the register-resident DAG kernel gave up its grid wait");
        }
        if (e->h_ctl->run_id != e->run_id) return fail(BN_ERR_HIP, "bp_dag kernel did not report (stale control block)");
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
static void report_abort_once(bn_engine* e, const char* what, int pause_runs) {
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

// ---- the one-launch execution paths of a single query: one driver per path ------------------------------------------------------
// wanted(): eligible AND chosen -- by the option ("small" / "mid" / "dag" / "multisweep": 0 never, 2 wherever eligible) or, at 1, by
// the defaults measured on 20 networks (scripts/time_paths.py, profiles/r05_paths.json).  run(): BN_OK, BN_ERR_STATE (a bounded wait
// gave up: the launch's workgroups were not all on the chip), or an error.  gave_up(): the path's bookkeeping of such an abort.
struct PathDriver {
    int id;                                             // bn_bp_last_path
    bool (*wanted)(const bn_engine*);
    int (*run)(bn_engine*, double eps, int32_t max_sweeps, double* copy_to);
    int (*gave_up)(bn_engine*);                         // BN_OK: go on with the next path
    void (*ran_ok)(bn_engine*);                         // may be null
    int32_t bn_engine::*cooldown;                       // runs left before the path is tried again
    bool reads_tile_evidence;                           // flush_evidence() first
};

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
static void resident_ran_ok(bn_engine* e) { e->resident_backoff = 8; }

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
static int small_gave_up(bn_engine*) { return BN_OK; }   // (one workgroup: it waits for nobody)

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

// ---- several evidence sets on one network (extension beside the drop-in: the reference's API takes one
// query at a time).  Resident-eligible networks run all sets in ONE launch that walks them round-robin
// (bn_resident.hip): one resident CPT serves every set and each set's barrier completes while the others
// compute.  Other networks run the sets one after another through the single-query path.  Either way
// every set's results are bit-identical to running it alone.
static int batch_reserve(bn_engine* e, int32_t n_sets) {
    bn_engine::Batch& bt = e->batch;
    if (n_sets <= bt.cap_sets) return BN_OK;
    const Plan& p = e->plan;
    HIPCHK(hipStreamSynchronize(e->stream));
    void* old[] = {bt.d_rec[0], bt.d_rec[1], bt.d_node[0], bt.d_node[1], bt.d_frozen, bt.d_beliefs, bt.d_res_hist, bt.d_sync, bt.d_ctl, bt.d_s_state,
                   bt.d_ev, bt.d_g_state, bt.d_g_frz, bt.d_g_sync};
    if (bt.h_ev) (void)hipHostFree(bt.h_ev);
    if (bt.h_beliefs) (void)hipHostFree(bt.h_beliefs);
    for (void* q : old)
        if (q) (void)hipFree(q);
    if (bt.h_ctl) (void)hipHostFree(bt.h_ctl);
    bt = bn_engine::Batch();
    int r;
    const size_t B = size_t(n_sets);
    for (int i = 0; i < 2; ++i) {
        if ((r = dalloc(&bt.d_rec[i], B * size_t(p.rec_total_doubles)))) return r;
        if ((r = dalloc(&bt.d_node[i], B * size_t(p.node_doubles)))) return r;
        HIPCHK(hipMemsetAsync(bt.d_rec[i], 0, std::max<size_t>(B * p.rec_total_doubles, 1) * 8, e->stream));
        HIPCHK(hipMemsetAsync(bt.d_node[i], 0, std::max<size_t>(B * p.node_doubles, 1) * 8, e->stream));
    }
    if ((r = dalloc(&bt.d_frozen, B * size_t(std::max(p.n_slots, 1))))) return r;
    if ((r = dalloc(&bt.d_beliefs, B * size_t(p.node_off[p.n])))) return r;
    if ((r = dalloc(&bt.d_res_hist, B * size_t(e->res_cap)))) return r;
    if ((r = dalloc(&bt.d_sync, std::min<size_t>(B, kResidentMaxSets)))) return r;
    if ((r = dalloc(&bt.d_ctl, B))) return r;
    if (e->small_ok && (r = dalloc(&bt.d_s_state, B * size_t(2 * e->small.M + 2 * e->small.N)))) return r;
    HIPCHK(hipMemsetAsync(bt.d_ctl, 0, sizeof(Ctl) * B, e->stream));  // done_run = 0: no run is marked done
    HIPCHK(hipMemsetAsync(bt.d_frozen, 0, B * size_t(std::max(p.n_slots, 1)), e->stream));
    HIPCHK(hipMemsetAsync(bt.d_beliefs, 0, std::max<size_t>(B * p.node_off[p.n], 1) * 8, e->stream));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&bt.h_ctl), sizeof(Ctl) * B, hipHostMallocMapped));
    std::memset(bt.h_ctl, 0, sizeof(Ctl) * B);
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&bt.h_ctl_dev), bt.h_ctl, 0));
    HIPCHK(hipStreamSynchronize(e->stream));
    bt.cap_sets = n_sets;
    return BN_OK;
}

// buffers of evidence set q inside the batch arrays
static BpBuffers batch_buffers_of(bn_engine* e, int32_t q) {
    const Plan& p = e->plan;
    const bn_engine::Batch& bt = e->batch;
    BpBuffers b = buffers_of(e);
    b.rec0 = bt.d_rec[0] + size_t(q) * p.rec_total_doubles;
    b.rec1 = bt.d_rec[1] + size_t(q) * p.rec_total_doubles;
    b.node0 = bt.d_node[0] + size_t(q) * p.node_doubles;
    b.node1 = bt.d_node[1] + size_t(q) * p.node_doubles;
    b.frozen = bt.d_frozen + size_t(q) * std::max(p.n_slots, 1);
    b.frozen_mark = 1;  // batches clear their marks with a memset per call
    b.beliefs = bt.d_beliefs + size_t(q) * p.node_off[p.n];
    b.res_hist = bt.d_res_hist + size_t(q) * e->res_cap;
    b.ctl = bt.d_ctl + q;
    return b;
}

// Batches want throughput; a layout built for the latency of one query (wide lane groups, any-arity tiles for
// nodes with many children: bn_plan.cpp) has up to 4x the wavefronts.  Such an engine answers batches of two or more
// sets through a second engine built from the same model with the dense layout (lanes_per_node = 2); the networks
// this concerns are small, so the second copy is too.  Created at the first such call.
static bn_engine* dense_engine_for_batch(bn_engine* e, int32_t n_sets, int& rc) {
    rc = BN_OK;
    if (!e->plan.latency_rules_applied || e->plan.nranks > 1 || n_sets < 2) return nullptr;
    if (e->small_ok && e->small_mode != 0 && e->multisweep != 0) return nullptr;  // one workgroup per set (bn_small.hip): the layout plays no part
    if (mid_applies(e)) return nullptr;                                           // ... or a few per set (bn_mid.hip)
    if (dag_applies(e)) return nullptr;                                           // ... or the register-resident DAG path, set by set (bn_dag.hip)
    if (!e->dense) {
        const Plan& p = e->plan;
        bn_model_desc d;
        d.n_nodes = p.n;
        d.k = p.k.data(); d.in_ptr = p.in_ptr.data(); d.in_idx = p.in_idx.data();
        d.cpt_off = p.cpt_off.data(); d.cpt = p.cpt_flat.data();
        d.device = e->device;
        d.lanes_per_node = p.group_wide ? 4 : 2;  // same lane-group split: same bits as this engine's single queries
        rc = bn_create(&d, &e->dense);
        if (rc) { e->dense = nullptr; return nullptr; }
    }
    e->dense->multisweep = e->multisweep;
    e->dense->small_mode = e->small_mode;
    e->dense->mid_mode = e->mid_mode;
    e->dense->dag_mode = e->dag_mode;
    e->dense->timing = e->timing;
    return e->dense;
}
static void adopt_batch_outcome(bn_engine* e) {  // what bn_bp_stats / bn_bp_last_path report after a forwarded batch
    e->last_path = e->dense->last_path;
    const bn_bp_stats own = e->stats;
    e->stats = e->dense->stats;
    e->stats.algorithmic_bytes_per_sweep = own.algorithmic_bytes_per_sweep;
    e->stats.layout_bytes_per_sweep = own.layout_bytes_per_sweep;
    e->stats.messages_per_sweep = own.messages_per_sweep;
}

// The batch's evidence (bt.d_ev) -> the sets' tile buffers: marks cleared, one bp_evidence_kernel per set.  No-op when done already.
static int flush_batch_evidence(bn_engine* e) {
    bn_engine::Batch& bt = e->batch;
    if (!bt.ev_deferred) return BN_OK;
    const Plan& p = e->plan;
    HIPCHK(hipMemsetAsync(bt.d_frozen, 0, size_t(bt.n_sets) * size_t(std::max(p.n_slots, 1)), e->stream));
    for (int32_t q = 0; q < bt.n_sets; ++q) {
        EvidenceArgs ea{batch_buffers_of(e, q), bt.ne[q], reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_node) + bt.ev_node_at[q],
                        reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_off) + bt.ev_off_at[q],
                        reinterpret_cast<double*>(bt.ev_base + bt.ev_b_val) + bt.ev_val_at[q]};
        if (int code = launch_bp_evidence(ea, e->stream))
            return fail(BN_ERR_HIP, std::string("bp_evidence launch failed: ") + hipGetErrorString(hipError_t(code)));
    }
    bt.ev_deferred = false;
    return BN_OK;
}

extern "C" int bn_bp_set_evidence_batch(bn_engine* e, int32_t n_sets, const int32_t* ne, const int32_t* ev_node,
                                        const int32_t* ev_off, const double* ev_val) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (e->host_only) return fail(BN_ERR_STATE, "engine was created with BN_DEVICE_HOST_ONLY: no GPU, no compute");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: bn_reload_cpt failed while uploading (destroy it and create a new one)");
    e->batch_on_dense = false;
    if (n_sets >= 1 && n_sets <= BN_MAX_BATCH_SETS) {
        int rc;
        if (bn_engine* de = dense_engine_for_batch(e, n_sets, rc)) {
            rc = bn_bp_set_evidence_batch(de, n_sets, ne, ev_node, ev_off, ev_val);
            e->batch_on_dense = rc == BN_OK;
            return rc;
        } else if (rc) {
            return rc;
        }
    }
    if (n_sets < 1 || n_sets > BN_MAX_BATCH_SETS) return fail(BN_ERR_ARG, "n_sets must be in 1.." + std::to_string(BN_MAX_BATCH_SETS));
    if (e->plan.nranks > 1) return fail(BN_ERR_STATE, "batched evidence sets are not available on sharded engines");
    if (!ne) return fail(BN_ERR_ARG, "null ne");
    const Plan& p = e->plan;
    // validate every set like bn_bp_set_evidence does; locate its slices of the concatenated arrays
    std::vector<int64_t> node_at(n_sets + 1, 0), off_at(n_sets + 1, 0), val_at(n_sets + 1, 0);
    for (int32_t q = 0; q < n_sets; ++q) {
        if (ne[q] < 0) return fail(BN_ERR_ARG, "negative evidence count");
        if (ne[q] > 0 && (!ev_node || !ev_off || !ev_val)) return fail(BN_ERR_ARG, "null evidence array");
        int rc = check_evidence(p, ne[q], ev_node ? ev_node + node_at[q] : nullptr, ev_off ? ev_off + off_at[q] : nullptr, e->ev_seen, e->ev_epoch);
        if (rc) return rc;
        node_at[q + 1] = node_at[q] + ne[q];
        off_at[q + 1] = off_at[q] + ne[q] + 1;
        val_at[q + 1] = val_at[q] + (ne[q] > 0 ? ev_off[off_at[q] + ne[q]] : 0);
    }
    ON_DEVICE(e);
    int rc = batch_reserve(e, n_sets);
    if (rc) return rc;
    bn_engine::Batch& bt = e->batch;
    bt.n_sets = n_sets;
    bt.have_run = false;
    bt.ne.assign(ne, ne + n_sets);
    bt.ev_node.assign(ev_node, ev_node + node_at[n_sets]);
    bt.ev_off.assign(ev_off, ev_off + (node_at[n_sets] > 0 || ev_off ? off_at[n_sets] : 0));
    bt.ev_val.assign(ev_val, ev_val + val_at[n_sets]);
    // one staging block [nodes | offs | vals | per-set meta]
    const size_t b_node = 0, b_off = size_t(node_at[n_sets]) * 4, b_val = (b_off + size_t(off_at[n_sets]) * 4 + 7) & ~size_t(7);
    const size_t b_meta = b_val + size_t(val_at[n_sets]) * 8;
    const size_t bytes = b_meta + size_t(n_sets) * 32;
    auto fill = [&](char* dst) {
        if (node_at[n_sets] > 0) {
            std::memcpy(dst + b_node, ev_node, size_t(node_at[n_sets]) * 4);
            std::memcpy(dst + b_val, ev_val, size_t(val_at[n_sets]) * 8);
        }
        if (ev_off) std::memcpy(dst + b_off, ev_off, size_t(off_at[n_sets]) * 4);
        int32_t* meta = reinterpret_cast<int32_t*>(dst + b_meta);  // per set {count, first node entry, first offset entry, first value, values}
        for (int32_t q = 0; q < n_sets; ++q) {
            meta[8 * q] = ne[q]; meta[8 * q + 1] = int32_t(node_at[q]); meta[8 * q + 2] = int32_t(off_at[q]); meta[8 * q + 3] = int32_t(val_at[q]);
            meta[8 * q + 4] = int32_t(val_at[q + 1] - val_at[q]); meta[8 * q + 5] = meta[8 * q + 6] = meta[8 * q + 7] = 0;
        }
    };
    bt.ev_b_node = b_node; bt.ev_b_off = b_off; bt.ev_b_val = b_val;
    bt.ev_node_at = node_at; bt.ev_off_at = off_at; bt.ev_val_at = val_at;
    bt.ev_deferred = true;
    bt.beliefs_on_host = false;
    if (e->small_ok || e->mid_ok || dag_applies(e)) {
        // Small networks: the block is page-locked host memory that the kernels read in place -- the one-workgroup kernel (one
        // workgroup per set) each set's arrays, no copy command, no evidence launch per set, no synchronisation here; the tile
        // buffers get the marks and vectors only if another path runs the batch (flush_batch_evidence).  (No kernel is in
        // flight when the block is rewritten: every run entry point synchronises before it returns.)
        if (bytes > bt.h_ev_cap) {
            if (bt.h_ev) (void)hipHostFree(bt.h_ev);
            bt.h_ev = nullptr;
            bt.h_ev_cap = std::max<size_t>(bytes * 2, 4096);
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&bt.h_ev), bt.h_ev_cap, hipHostMallocMapped));
            HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&bt.ev_base), bt.h_ev, 0));
        }
        fill(bt.h_ev);
        bt.d_ev_meta = reinterpret_cast<int32_t*>(bt.ev_base + b_meta);
        return BN_OK;
    }
    // every other network: one H2D copy, then one evidence kernel per set
    if (bytes > bt.ev_cap) {
        if (bt.d_ev) (void)hipFree(bt.d_ev);
        bt.d_ev = nullptr;
        bt.ev_cap = std::max<size_t>(bytes * 2, 4096);
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&bt.d_ev), bt.ev_cap));
    }
    bt.ev_base = bt.d_ev;
    std::vector<char> host(std::max<size_t>(bytes, 1));
    fill(host.data());
    HIPCHK(hipMemcpyAsync(bt.d_ev, host.data(), bytes, hipMemcpyHostToDevice, e->stream));
    bt.d_ev_meta = reinterpret_cast<int32_t*>(bt.ev_base + b_meta);
    if ((rc = flush_batch_evidence(e))) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));  // `host` is a local
    return BN_OK;
}

// sets [first, first + count) through the resident kernel, round-robin in one launch (count <= kResidentMaxSets)
static int run_batch_resident_chunk(bn_engine* e, double eps, int32_t max_sweeps, int32_t first, int32_t count, int32_t& launches,
                                    float& ms, double& dev_ticks) {
    bn_engine::Batch& bt = e->batch;
    const Plan& p = e->plan;
    hipStream_t s = e->stream;
    int32_t begin = 0;
    uint32_t mask = (1u << count) - 1u;
    for (;;) {
        if (bt.sync_dirty || bt.gen_base > (1u << 29)) {
            HIPCHK(hipMemsetAsync(bt.d_sync, 0, sizeof(ResidentSync) * size_t(std::min(bt.cap_sets, kResidentMaxSets)), s));
            bt.sync_dirty = false;
            bt.gen_base = 0;
        }
        *e->h_abort = 0;
        ResidentArgs a{batch_buffers_of(e, first), eps, max_sweeps, begin, kResidentBudget, e->run_id, bt.gen_base, 5000000ull, bt.d_sync,
                       bt.h_ctl_dev + first, e->grid_resident, e->resident_waves, count, mask, p.rec_total_doubles, p.node_doubles,
                       int64_t(std::max(p.n_slots, 1)), p.node_off[p.n], e->res_cap, nullptr, nullptr, nullptr, 0, nullptr, 1, 0, e->h_abort_dev};
        if (e->timing) {
            int rc = ensure_events(e, 2);
            if (rc) return rc;
            HIPCHK(hipEventRecord(e->events[0], s));
        }
        if (int code = launch_bp_resident(a, e->grid_resident + resident_service_blocks(e->grid_resident), e->resident_lean, s))
            return fail(BN_ERR_HIP, std::string("bp_resident launch failed: ") + hipGetErrorString(hipError_t(code)));
        if (e->timing) HIPCHK(hipEventRecord(e->events[1], s));
        HIPCHK(hipStreamSynchronize(s));
        ++launches;
        if (e->timing) {
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, e->events[0], e->events[1]));
            ms += t;
        }
        bt.gen_base += kResidentBudget + 1;
        if (*e->h_abort != 0) {
            bt.sync_dirty = true;
            return fail(BN_ERR_STATE, "resident kernel gave up a barrier wait");
        }
        uint32_t next = 0;
        for (int32_t q = 0; q < count; ++q) {
            if (!((mask >> q) & 1u)) continue;
            const Ctl& c = bt.h_ctl[first + q];
            if (c.run_id != e->run_id || c.done < 0) bt.sync_dirty = true;
            if (c.run_id != e->run_id) return fail(BN_ERR_STATE, "resident kernel did not report (stale control block)");
            if (c.done < 0) return fail(BN_ERR_STATE, "resident kernel gave up a barrier wait");
            bt.sweeps[first + q] = c.n_sweeps;
            bt.residual[first + q] = c.last_res;
            if (c.done == 0) next |= 1u << q;
        }
        dev_ticks += double(bt.h_ctl[first].t_last - bt.h_ctl[first].t_first);
        if (next == 0) break;
        mask = next;
        begin += kResidentBudget;
    }
    return BN_OK;
}

// every set through the resident kernel: up to kResidentMaxSets per launch, further sets in further launches
static int run_batch_resident(bn_engine* e, double eps, int32_t max_sweeps) {
    bn_engine::Batch& bt = e->batch;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    int32_t launches = 0;
    double dev_ticks = 0.0;
    float ms = 0.f;
    const int32_t chunks = (bt.n_sets + kResidentMaxSets - 1) / kResidentMaxSets;
    for (int32_t c = 0, first = 0; c < chunks; ++c) {
        const int32_t count = (bt.n_sets - first + (chunks - c) - 1) / (chunks - c);  // balanced chunk sizes
        int rc = run_batch_resident_chunk(e, eps, max_sweeps, first, count, launches, ms, dev_ticks);
        if (rc) return rc;
        first += count;
    }
    e->last_path = 2;
    e->stats.sweep_launches = launches;
    e->stats.sweep_kernel_ms = ms;
    e->stats.sweep_devclock_ms = float(dev_ticks * 1e-5);
    e->stats.sweeps = *std::max_element(bt.sweeps.begin(), bt.sweeps.end());
    return BN_OK;
}

// Every set in each per-sweep launch (blockIdx.y = evidence set): any tile variants.  The sets share the launch
// and its latency -- what a small or latency-bound network pays for -- and the CPT lines in the caches; each keeps
// its own records, node vectors, marks, residual slots and done mark, so it stops on the sweep its single run
// stops on (a converged set's blocks return at once in the launches the others still need).
static int run_batch_launches(bn_engine* e, double eps, int32_t max_sweeps) {
    bn_engine::Batch& bt = e->batch;
    const Plan& p = e->plan;
    hipStream_t s = e->stream;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    const int32_t B = bt.n_sets;
    if (!bt.rows_clean) {  // an earlier batched run did not end through its finish kernel
        for (int32_t q = 0; q < bt.cap_sets; ++q)
            if (int code = launch_bp_reset(batch_buffers_of(e, q), s))
                return fail(BN_ERR_HIP, std::string("bp_reset launch failed: ") + hipGetErrorString(hipError_t(code)));
    }
    bt.rows_clean = false;
    const SetStrides st{p.rec_total_doubles, p.node_doubles, int64_t(std::max(p.n_slots, 1)), p.node_off[p.n], e->res_cap};
    const BpBuffers b0 = batch_buffers_of(e, 0);
    const int32_t nt = int32_t(p.tiles.size());
    const int grid = ((nt + 1 + kWavesPerBlock - 1) / kWavesPerBlock + 7) & ~7;
    static const bool no_light = std::getenv("BN_NO_LIGHT") != nullptr;
    int32_t launched = 0;
    int32_t batch = bt.predicted_sweeps > 0 ? bt.predicted_sweeps : (e->predicted_sweeps > 0 ? e->predicted_sweeps : 8);
    for (;;) {
        if (max_sweeps > 0) batch = std::min(batch, max_sweeps - launched);
        for (int32_t i = 0; i < batch; ++i) {
            const int32_t sweep = launched + i;
            const int cur = sweep & 1;
            SweepArgs sa{b0, bt.d_rec[cur], bt.d_rec[cur ^ 1], bt.d_node[cur], bt.d_node[cur ^ 1], eps, sweep, 0, nt, 1, e->run_id, st};
            // B == 1 runs the plain instantiation on set 0's buffers
            if (launch_bp_sweep(sa, grid, B, false, p.light && !no_light, p.variants, s)) return fail(BN_ERR_HIP, "bp_sweep launch failed");
        }
        launched += batch;
        FinishArgs fa{b0, eps, launched, (max_sweeps > 0 && launched >= max_sweeps) ? 1 : 0, e->run_id, bt.h_ctl_dev, st};
        if (launch_bp_finish(fa, e->grid_tiles, B, s)) return fail(BN_ERR_HIP, "bp_finish launch failed");
        HIPCHK(hipStreamSynchronize(s));
        bool all_done = true;
        for (int32_t q = 0; q < B; ++q) {
            if (bt.h_ctl[q].run_id != e->run_id) return fail(BN_ERR_STATE, "finish kernel did not report (stale control block)");
            if (bt.h_ctl[q].done == 0) all_done = false;
        }
        if (all_done) break;
        batch = 8;
    }
    bt.rows_clean = true;  // every set's run ended in a finish kernel that saw it over
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int32_t q = 0; q < B; ++q) {
        bt.sweeps[q] = bt.h_ctl[q].n_sweeps;
        bt.residual[q] = bt.h_ctl[q].last_res;
        t0 = std::min(t0, bt.h_ctl[q].t_first);
        t1 = std::max(t1, bt.h_ctl[q].t_last);
    }
    bt.predicted_sweeps = *std::max_element(bt.sweeps.begin(), bt.sweeps.end());
    e->last_path = 0;
    e->stats.sweep_launches = launched;
    e->stats.sweep_kernel_ms = 0.f;
    e->stats.sweep_devclock_ms = t1 > t0 ? float(double(t1 - t0) * 1e-5) : 0.f;
    e->stats.sweeps = bt.predicted_sweeps;
    return BN_OK;
}

// Small networks: one workgroup per evidence set, all sets in ONE launch, each set stopping by itself (bn_small.hip).
static int run_batch_small(bn_engine* e, double eps, int32_t max_sweeps) {
    bn_engine::Batch& bt = e->batch;
    const Plan& p = e->plan;
    hipStream_t s = e->stream;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    const int32_t B = bt.n_sets;
    const SetStrides st{p.rec_total_doubles, p.node_doubles, int64_t(std::max(p.n_slots, 1)), p.node_off[p.n], e->res_cap};
    const int64_t state_stride = 2 * int64_t(e->small.M) + 2 * int64_t(e->small.N);
    SmallArgs a = small_args_of(e, batch_buffers_of(e, 0), eps, max_sweeps, 0, bt.h_ctl_dev);
    a.state = bt.d_s_state; a.sets = st; a.state_stride = state_stride;
    const size_t per_set = size_t(p.node_off[p.n]);
    if (bt.direct_out) {  // bn_bp_run_batch: the marginals go straight into page-locked host memory (no copy command, no second sync)
        if (size_t(B) * per_set > bt.h_beliefs_cap) {
            if (bt.h_beliefs) (void)hipHostFree(bt.h_beliefs);
            bt.h_beliefs = nullptr;
            bt.h_beliefs_cap = size_t(bt.cap_sets) * per_set;
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&bt.h_beliefs), std::max<size_t>(bt.h_beliefs_cap, 1) * sizeof(double), hipHostMallocMapped));
            HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&bt.h_beliefs_dev), bt.h_beliefs, 0));
        }
        a.b.beliefs = bt.h_beliefs_dev;
    }
    bt.beliefs_on_host = bt.direct_out;
    auto evidence_of = [&](SmallArgs& x, bool per_set_meta, int32_t q) {
        if (!bt.ev_deferred) return;  // the tile buffers hold it
        x.ev_mode = 1;
        x.ev_node = reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_node);
        x.ev_off = reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_off);
        x.ev_val = reinterpret_cast<double*>(bt.ev_base + bt.ev_b_val);
        x.ev_meta = per_set_meta ? bt.d_ev_meta : bt.d_ev_meta + 8 * q;  // (a single-set launch reads entry `blockIdx.x` = 0)
    };
    evidence_of(a, true, 0);
    if (int code = launch_bp_small(a, e->small.waves, e->small.lds_bytes, B, s))
        return fail(BN_ERR_HIP, std::string("bp_small launch failed: ") + hipGetErrorString(hipError_t(code)));
    HIPCHK(hipStreamSynchronize(s));
    int32_t launches = 1;
    for (int32_t q = 0; q < B; ++q) {
        if (bt.h_ctl[q].run_id != e->run_id) return fail(BN_ERR_HIP, "bp_small kernel did not report (stale control block)");
        while (bt.h_ctl[q].done == 0) {  // a set that used up the launch's budget of iterations goes on by itself
            SmallArgs c = small_args_of(e, batch_buffers_of(e, q), eps, max_sweeps, bt.h_ctl[q].n_sweeps, bt.h_ctl_dev + q);
            c.state = bt.d_s_state + size_t(q) * state_stride;
            if (bt.direct_out) c.b.beliefs = bt.h_beliefs_dev + size_t(q) * per_set;
            evidence_of(c, false, q);
            if (int code = launch_bp_small(c, e->small.waves, e->small.lds_bytes, 1, s))
                return fail(BN_ERR_HIP, std::string("bp_small launch failed: ") + hipGetErrorString(hipError_t(code)));
            HIPCHK(hipStreamSynchronize(s));
            ++launches;
        }
    }
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int32_t q = 0; q < B; ++q) {
        bt.sweeps[q] = bt.h_ctl[q].n_sweeps;
        bt.residual[q] = bt.h_ctl[q].last_res;
        t0 = std::min(t0, bt.h_ctl[q].t_first);
        t1 = std::max(t1, bt.h_ctl[q].t_last);
    }
    bt.predicted_sweeps = *std::max_element(bt.sweeps.begin(), bt.sweeps.end());
    e->last_path = 3;
    e->stats.sweep_launches = launches;
    e->stats.sweep_kernel_ms = 0.f;
    e->stats.sweep_devclock_ms = t1 > t0 ? float(double(t1 - t0) * 1e-5) : 0.f;
    e->stats.sweeps = bt.predicted_sweeps;
    return BN_OK;
}

// Mid-size networks: every set runs exactly like a single query (same kernel, same bits), as many sets per launch as fit the
// chip with a workgroup per CU (the grid barrier needs every workgroup of a set resident).  BN_ERR_STATE: a grid wait gave up.
static int run_batch_mid(bn_engine* e, double eps, int32_t max_sweeps) {
    bn_engine::Batch& bt = e->batch;
    const Plan& p = e->plan;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    const int32_t B = bt.n_sets, nparts = int32_t(e->mid.parts.size());
    const int32_t per_launch = std::max(1, std::min(B, (e->n_cus * 9 / 10) / nparts));
    int rc;
    if ((rc = mid_reserve_slots(e, per_launch))) return rc;
    const SetStrides st{p.rec_total_doubles, p.node_doubles, int64_t(std::max(p.n_slots, 1)), p.node_off[p.n], e->res_cap};
    const BpBuffers b0 = batch_buffers_of(e, 0);
    auto evidence_of = [&](MidArgs& x) {
        if (!bt.ev_deferred) return;  // the tile buffers hold it
        x.ev_mode = 1;
        x.ev_node = reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_node);
        x.ev_off = reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_off);
        x.ev_val = reinterpret_cast<double*>(bt.ev_base + bt.ev_b_val);
        x.ev_meta = bt.d_ev_meta;
    };
    int32_t launches = 0;
    for (int32_t first = 0; first < B; first += per_launch) {
        const int32_t count = std::min(per_launch, B - first);
        MidArgs a = mid_args_of(e, b0, st, bt.h_ctl_dev, eps, max_sweeps, 0, first, 0);
        evidence_of(a);
        if ((rc = mid_launch(e, a, count, nullptr, nullptr))) return rc;
        ++launches;
        for (int32_t q = first; q < first + count; ++q) {
            if (bt.h_ctl[q].done < 0) return fail(BN_ERR_STATE, "a workgroup of the mid-size kernel gave up its grid wait");
            if (bt.h_ctl[q].run_id != e->run_id) return fail(BN_ERR_HIP, "bp_mid kernel did not report (stale control block)");
            while (bt.h_ctl[q].done == 0) {  // a set that used up the launch's budget of iterations goes on by itself, in its slot
                MidArgs c = mid_args_of(e, b0, st, bt.h_ctl_dev, eps, max_sweeps, bt.h_ctl[q].n_sweeps, q, q - first);
                evidence_of(c);
                if ((rc = mid_launch(e, c, 1, nullptr, nullptr))) return rc;
                ++launches;
                if (bt.h_ctl[q].done < 0) return fail(BN_ERR_STATE, "a workgroup of the mid-size kernel gave up its grid wait");
            }
        }
    }
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int32_t q = 0; q < B; ++q) {
        bt.sweeps[q] = bt.h_ctl[q].n_sweeps;
        bt.residual[q] = bt.h_ctl[q].last_res;
        t0 = std::min(t0, bt.h_ctl[q].t_first);
        t1 = std::max(t1, bt.h_ctl[q].t_last);
    }
    bt.predicted_sweeps = *std::max_element(bt.sweeps.begin(), bt.sweeps.end());
    e->last_path = 4;
    e->stats.sweep_launches = launches;
    e->stats.sweep_kernel_ms = 0.f;
    e->stats.sweep_devclock_ms = t1 > t0 ? float(double(t1 - t0) * 1e-5) : 0.f;
    e->stats.sweeps = bt.predicted_sweeps;
    return BN_OK;
}

// The register-resident DAG path (bn_dag.hip) answers a batch one set after another: every set is a single query's launch -- the
// same kernel, the same bits -- reading its evidence from the batch's staging block and writing its marginals and residual history
// into the set's slots.  BN_ERR_STATE: a grid wait gave up.
// Sets [first, first + count) of the batch in ONE launch of the register-resident DAG kernel: the sets take turns inside an
// iteration, so a set's barrier completes while the others sweep, and one set of CPT registers serves them all (bn_dag.hip,
// dag_drive).  Every set has its own state, marks, barrier words, residual history and control block and keeps the bits and the
// sweep count of its single run.  left[q] = true: set q did not finish here (more than kDagBudget sweeps) and is run on its own.
// BN_ERR_STATE: a grid wait gave up.
// (enqueue only: the chunks of a batch follow each other on the stream -- the next chunk's evidence lands in the state slots when the
// previous chunk's kernel has left them -- and the host waits once, for all of them: collect_batch_dag_chunk reads the outcome.)
struct DagChunk { int32_t first, count; uint32_t run_id; };
static int enqueue_batch_dag_chunk(bn_engine* e, double eps, int32_t max_sweeps, int32_t first, int32_t count, DagChunk& chunk) {
    bn_engine::Batch& bt = e->batch;
    const Plan& p = e->plan;
    const DagPlan& dp = e->dag;
    hipStream_t s = e->stream;
    const size_t state_d = size_t(dag_state_doubles(dp.E, dp.n));
    if (bt.dag_sets < kDagMaxSets) {   // first use: every set's state, marks and barrier words
        int r;
        if ((r = dalloc(&bt.d_g_state, state_d * kDagMaxSets))) return r;
        if ((r = dalloc(&bt.d_g_frz, size_t(dp.n) * kDagMaxSets))) return r;
        if ((r = dalloc(&bt.d_g_sync, size_t(kDagMaxSets)))) return r;
        HIPCHK(hipMemsetAsync(bt.d_g_state, 0, state_d * kDagMaxSets * sizeof(double), s));
        HIPCHK(hipMemsetAsync(bt.d_g_frz, 0, size_t(dp.n) * kDagMaxSets, s));
        bt.dag_sets = kDagMaxSets;
        bt.dag_mark = 0;
        bt.dag_sync_dirty = true;
    }
    if (bt.dag_mark == 255) {  // the mark values are used up: start over
        HIPCHK(hipMemsetAsync(bt.d_g_frz, 0, size_t(dp.n) * kDagMaxSets, s));
        bt.dag_mark = 0;
    }
    ++bt.dag_mark;
    {   // pi(v) = lambda(v) = the given vector in both buffers, node marked (:68-73): every set of the chunk in one launch
        DagEvidenceBatch eb{};
        DagInitBatch ib{};
        for (int32_t q = 0; q < count; ++q) {
            const int32_t g = first + q;
            eb.set[q] = DagEvidenceArgs{bt.ne[g], dp.n, dp.E, reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_node) + bt.ev_node_at[g],
                                        reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_off) + bt.ev_off_at[g],
                                        reinterpret_cast<double*>(bt.ev_base + bt.ev_b_val) + bt.ev_val_at[g], bt.d_g_state + size_t(q) * state_d,
                                        bt.d_g_frz + size_t(q) * dp.n, bt.dag_mark, e->d_g_k};
            ib.set[q] = DagInitArgs{dp.n, dp.E, e->d_g_inptr, e->d_g_inidx, e->d_g_k, e->d_g_init, bt.d_g_state + size_t(q) * state_d,
                                    bt.d_g_frz + size_t(q) * dp.n, bt.dag_mark};
        }
        if (int code = launch_dag_evidence_batch(eb, count, s))
            return fail(BN_ERR_HIP, std::string("dag_evidence launch failed: ") + hipGetErrorString(hipError_t(code)));
        if (!dp.uniform4) {
            if (int code = launch_dag_init_batch(ib, count, s))
                return fail(BN_ERR_HIP, std::string("dag_init launch failed: ") + hipGetErrorString(hipError_t(code)));
        }
    }
    if (bt.dag_sync_dirty || bt.dag_gen_base > (1u << 29)) {
        HIPCHK(hipMemsetAsync(bt.d_g_sync, 0, sizeof(ResidentSync) * size_t(kDagMaxSets), s));
        bt.dag_sync_dirty = false;
        bt.dag_gen_base = 0;
    }
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    chunk = DagChunk{first, count, e->run_id};
    DagArgs a{};
    a.b = buffers_of(e);
    a.b.beliefs = bt.d_beliefs + size_t(first) * p.node_off[p.n];
    a.b.res_hist = bt.d_res_hist + size_t(first) * e->res_cap;
    a.eps = eps; a.max_sweeps = max_sweeps; a.sweep_begin = 0; a.budget = kDagBudget; a.run_id = e->run_id;
    a.gen_base = bt.dag_gen_base;
    a.timeout_ticks = 5000000ull;
    a.sync = bt.d_g_sync; a.host_ctl = bt.h_ctl_dev + first; a.host_abort = e->h_abort_dev;
    a.n = dp.n; a.E = dp.E; a.n_blocks = dp.blocks;
    a.tiles = e->d_g_tiles; a.slot_ptr = e->d_g_slotptr; a.cnode = e->d_g_cnode; a.pitem = e->d_g_pitem; a.oedge = e->d_g_oedge;
    a.cpt_img = e->d_g_cpt; a.npi_init = e->d_g_init; a.state = bt.d_g_state; a.frz = bt.d_g_frz; a.frz_mark = bt.dag_mark;
    static const int poll_sleep = std::getenv("BN_DAG_SLEEP") ? std::atoi(std::getenv("BN_DAG_SLEEP")) : 1;
    static const int first_delay = std::getenv("BN_DAG_DELAY") ? std::atoi(std::getenv("BN_DAG_DELAY")) : 30;
    a.poll_sleep = poll_sleep;
    a.first_poll_delay = first_delay;
    a.n_sets = count; a.set_mask = (1u << count) - 1u;
    a.state_init = dp.uniform4 ? 0 : 1; a.node_k = e->d_g_k; a.node_off = e->d_g_noff;
    a.state_stride = int64_t(state_d); a.frz_stride = dp.n; a.belief_stride = p.node_off[p.n]; a.res_hist_stride = e->res_cap;
    for (int32_t q = 0; q < count; ++q) bt.h_ctl[first + q].run_id = 0;
    if (int code = launch_bp_dag(a, dp.stream, s))
        return fail(BN_ERR_HIP, std::string("bp_dag launch failed: ") + hipGetErrorString(hipError_t(code)));
    bt.dag_gen_base += kDagBudget + 1;
    return BN_OK;
}

// after the stream has drained.  left[q] = true: set q did not finish in its launch (more than kDagBudget sweeps) and is run on its own.
static int collect_batch_dag_chunk(bn_engine* e, const DagChunk& chunk, std::vector<char>& left, double& dev_ms, int32_t& max_sw) {
    bn_engine::Batch& bt = e->batch;
    const int32_t first = chunk.first, count = chunk.count;
    bool gave_up = *e->h_abort != 0, stale = false;
    for (int32_t q = 0; q < count; ++q) {
        gave_up = gave_up || bt.h_ctl[first + q].done < 0;
        stale = stale || bt.h_ctl[first + q].run_id != chunk.run_id;
    }
    if (gave_up || stale) bt.dag_sync_dirty = true;
    if (gave_up) return fail(BN_ERR_STATE, "a block of the register-resident DAG kernel gave up its grid wait");
    if (stale) return fail(BN_ERR_HIP, "bp_dag kernel did not report (stale control block)");
    dev_ms += double(bt.h_ctl[first].t_last - bt.h_ctl[first].t_first) * 1e-5;
    for (int32_t q = 0; q < count; ++q) {
        const Ctl& c = bt.h_ctl[first + q];
        if (c.done == 0) { left[first + q] = 1; continue; }   // the budget of one launch ran out: this set goes on alone
        bt.sweeps[first + q] = c.n_sweeps;
        bt.residual[first + q] = c.last_res;
        max_sw = std::max(max_sw, c.n_sweeps);
    }
    return BN_OK;
}

static int run_batch_dag(bn_engine* e, double eps, int32_t max_sweeps) {
    bn_engine::Batch& bt = e->batch;
    const Plan& p = e->plan;
    int rc = BN_OK;
    int32_t launches = 0, max_sw = 0;
    double dev_ms = 0.0;
    std::vector<char> left(size_t(bt.n_sets), 0);
    // how many sets share a launch (BN_DAG_SETS, default 8; 1 = one after another).  Config 2, us per set-sweep at B = 16: 8.7 / 6.9 / 6.2 / 5.9
    // with 1 / 2 / 4 / 8 sets per launch (scripts/time_dag_batch.py)
    static const int per_launch = std::max(1, std::min(kDagMaxSets, std::getenv("BN_DAG_SETS") ? std::atoi(std::getenv("BN_DAG_SETS")) : kDagMaxSets));
    if (per_launch > 1 && bt.n_sets > 1) {
        std::vector<DagChunk> chunks;
        *e->h_abort = 0;
        for (int32_t first = 0; first < bt.n_sets && rc == BN_OK; first += per_launch) {
            chunks.emplace_back();
            rc = enqueue_batch_dag_chunk(e, eps, max_sweeps, first, std::min(per_launch, bt.n_sets - first), chunks.back());
            if (rc != BN_OK) chunks.pop_back();
        }
        // (also after a failed enqueue: what is on the stream writes into the batch's buffers)
        const hipError_t drained = hipStreamSynchronize(e->stream);
        if (drained != hipSuccess && rc == BN_OK) rc = fail(BN_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(drained));
        launches += int32_t(chunks.size());
        for (const DagChunk& c : chunks) {
            const int rc_c = collect_batch_dag_chunk(e, c, left, dev_ms, max_sw);
            if (rc == BN_OK) rc = rc_c;
        }
        if (*e->h_abort != 0) { *e->h_abort = 0; bt.dag_sync_dirty = true; }
        if (rc != BN_OK) return rc;
    } else {
        std::fill(left.begin(), left.end(), 1);
    }
    // sets left over (a run beyond one launch's budget; a batch of one): through the single-query path, one after another
    const int32_t keep_ne = e->ev_ne;
    int32_t* const keep_node = e->d_ev_node;
    int32_t* const keep_off = e->d_ev_off;
    double* const keep_val = e->d_ev_val;
    double* const keep_override = e->beliefs_override;
    bool any_left = false;
    for (int32_t q = 0; q < bt.n_sets && rc == BN_OK; ++q) {
        if (!left[q]) continue;
        any_left = true;
        e->ev_ne = bt.ne[q];
        e->d_ev_node = reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_node) + bt.ev_node_at[q];
        e->d_ev_off = reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_off) + bt.ev_off_at[q];
        e->d_ev_val = reinterpret_cast<double*>(bt.ev_base + bt.ev_b_val) + bt.ev_val_at[q];
        e->dag_ev_applied = false;
        e->beliefs_override = bt.d_beliefs + size_t(q) * p.node_off[p.n];
        rc = run_dag(e, eps, max_sweeps, nullptr);
        if (rc != BN_OK) break;
        bt.sweeps[q] = e->last_ctl.n_sweeps;
        bt.residual[q] = e->last_ctl.last_res;
        const int32_t cnt = std::min(e->last_ctl.n_sweeps, e->res_cap);
        if (cnt > 0)
            HIPCHK(hipMemcpyAsync(bt.d_res_hist + size_t(q) * e->res_cap, e->d_res_hist, sizeof(double) * cnt, hipMemcpyDeviceToDevice, e->stream));
        launches += e->stats.sweep_launches;
        dev_ms += e->stats.sweep_devclock_ms;
        max_sw = std::max(max_sw, e->last_ctl.n_sweeps);
    }
    if (any_left) {
        // the single-query evidence in force is what the engine's own staging block holds: applied again at its next run
        e->ev_ne = keep_ne; e->d_ev_node = keep_node; e->d_ev_off = keep_off; e->d_ev_val = keep_val;
        e->dag_ev_applied = false;
        e->beliefs_override = keep_override;
    }
    if (rc != BN_OK) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    bt.predicted_sweeps = max_sw;
    e->last_path = 5;
    e->stats.sweep_launches = launches;
    e->stats.sweep_kernel_ms = 0.f;
    e->stats.sweep_devclock_ms = float(dev_ms);
    e->stats.sweeps = max_sw;
    return BN_OK;
}

// ---- the one-launch paths of a batch (bn_bp_run_batch_device): the PathDriver table of single queries, batch forms ----------------
// Which way a batch goes (measured, scripts/time_batch.py, us per set-sweep at the best batch size of either path): a small network runs one
// workgroup per set; otherwise the register-resident DAG path and the several-workgroup item kernel where their single-query policy
// chooses them; the resident tiles from ~900 tiles up -- per-sweep launches with one set per blockIdx.y share the launch latency among
// the sets, which is what smaller networks pay for (128x128 grid: 1.8 vs 7.7 resident, 200x200: 5.0 vs 8.0); on larger ones the CPT
// traffic the resident kernel saves weighs more (250x250: 9.1 vs 8.2, 316x316: 14.6 vs 8.6).  "multisweep" 2 forces the resident
// kernel wherever eligible, 0 the launches; "dag" 2 puts the DAG path in front of the one-workgroup path, as for single queries.
static bool batch_small_wanted(const bn_engine* e) {
    return e->small_ok && e->small_mode != 0 && e->multisweep != 0 && e->batch.d_s_state != nullptr && !(e->dag_mode == 2 && e->dag_ok);
}
static bool batch_dag_wanted(const bn_engine* e) { return !batch_small_wanted(e) && dag_applies(e) && e->batch.ev_base != nullptr && e->plan.nranks == 1; }
static bool batch_mid_wanted(const bn_engine* e) { return !batch_small_wanted(e) && mid_applies(e) && e->batch.ev_base != nullptr; }
static bool batch_resident_wanted(const bn_engine* e) {
    constexpr int64_t kResidentBatchMinTiles = 900;
    if (batch_small_wanted(e)) return false;
    return e->resident_ok && (e->multisweep == 2 || (e->multisweep == 1 && int64_t(e->plan.tiles.size()) >= kResidentBatchMinTiles));
}
static int run_batch_small_d(bn_engine* e, double eps, int32_t max_sweeps, double*) { return run_batch_small(e, eps, max_sweeps); }
static int run_batch_dag_d(bn_engine* e, double eps, int32_t max_sweeps, double*) { return run_batch_dag(e, eps, max_sweeps); }
static int run_batch_mid_d(bn_engine* e, double eps, int32_t max_sweeps, double*) { return run_batch_mid(e, eps, max_sweeps); }
static int run_batch_resident_d(bn_engine* e, double eps, int32_t max_sweeps, double*) { return run_batch_resident(e, eps, max_sweeps); }
static int batch_dag_gave_up(bn_engine* e) {
    ++e->dag_aborts;
    e->dag_cooldown = 64;
    report_abort_once(e, "the register-resident DAG kernel (bn_dag.hip, batch)", 64);
    return BN_OK;
}
static int batch_mid_gave_up(bn_engine* e) {
    ++e->mid_aborts;
    e->mid_cooldown = 64;
    report_abort_once(e, "the several-workgroup item kernel (bn_mid.hip, batch)", 64);
    return BN_OK;
}
static int batch_resident_gave_up(bn_engine* e) {
    ++e->resident_aborts;
    e->resident_cooldown = e->resident_backoff;
    e->resident_backoff = std::min(e->resident_backoff * 2, 1024);
    report_abort_once(e, "the resident-tile kernel (bn_resident.hip, batch)", e->resident_cooldown);
    return BN_OK;
}
static const PathDriver kBatchPaths[] = {
    {3, batch_small_wanted, run_batch_small_d, small_gave_up, nullptr, &bn_engine::small_cooldown, false},
    {5, batch_dag_wanted, run_batch_dag_d, batch_dag_gave_up, nullptr, &bn_engine::dag_cooldown, false},
    {4, batch_mid_wanted, run_batch_mid_d, batch_mid_gave_up, nullptr, &bn_engine::mid_cooldown, false},
    {2, batch_resident_wanted, run_batch_resident_d, batch_resident_gave_up, resident_ran_ok, &bn_engine::resident_cooldown, true},
};

extern "C" int bn_bp_run_batch_device(bn_engine* e, double eps, int32_t max_sweeps, int32_t* sweeps_out, double* residual_out) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (e->batch_on_dense && e->dense) {
        e->dense->multisweep = e->multisweep;
        e->dense->small_mode = e->small_mode;
        e->dense->mid_mode = e->mid_mode;
        e->dense->dag_mode = e->dag_mode;
        const int rc = bn_bp_run_batch_device(e->dense, eps, max_sweeps, sweeps_out, residual_out);
        if (rc == BN_OK) adopt_batch_outcome(e);
        return rc;
    }
    if (e->host_only) return fail(BN_ERR_STATE, "engine was created with BN_DEVICE_HOST_ONLY: no GPU, no compute");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: bn_reload_cpt failed while uploading (destroy it and create a new one)");
    if (max_sweeps < 0) return fail(BN_ERR_ARG, "max_sweeps < 0");
    bn_engine::Batch& bt = e->batch;
    if (bt.n_sets < 1) return fail(BN_ERR_STATE, "call bn_bp_set_evidence_batch first");
    const auto t_begin = std::chrono::steady_clock::now();
    ON_DEVICE(e);
    bt.sweeps.assign(bt.n_sets, 0);
    bt.residual.assign(bt.n_sets, 0.0);
    bt.beliefs_on_host = false;
    // The one-launch paths of a batch, in the order of kBatchPaths (the same drivers' table as a single query's, with the batch forms of
    // wanted / run): the first that wants the batch and is not paused runs every set; one that gives up a bounded wait pauses itself,
    // and the whole batch is run again by the next; what none of them takes runs with one launch per sweep, one set per blockIdx.y.
    int rc = BN_ERR_STATE;
    bool evidence_flushed = false, restage = false;
    for (const PathDriver& d : kBatchPaths) {
        if (!d.wanted(e)) continue;
        if (d.reads_tile_evidence && !evidence_flushed) {   // the tile kernels read the sets' evidence from their own buffers
            if ((rc = flush_batch_evidence(e))) return rc;
            evidence_flushed = true;
        }
        int32_t& cooldown = e->*(d.cooldown);
        if (cooldown > 0) { --cooldown; rc = BN_ERR_STATE; continue; }   // paused after a launch that gave up
        rc = d.run(e, eps, max_sweeps, nullptr);
        if (rc == BN_OK) {
            if (d.ran_ok) d.ran_ok(e);
            break;
        }
        if (rc != BN_ERR_STATE) return rc;
        if (int g = d.gave_up(e)) return g;   // counters, pause, one line on stderr
        bt.sweeps.assign(bt.n_sets, 0);       // the whole batch again on the next path
        bt.residual.assign(bt.n_sets, 0.0);
        restage = restage || d.reads_tile_evidence;
    }
    if (rc != BN_OK) {
        if (restage) {
            // marks / vectors possibly half-written by an aborted launch of the tile kernels: apply every set's evidence again
            std::vector<int32_t> ne = bt.ne, ev_node = bt.ev_node, ev_off = bt.ev_off;
            std::vector<double> ev_val = bt.ev_val;
            rc = bn_bp_set_evidence_batch(e, int32_t(ne.size()), ne.data(), ev_node.data(), ev_off.data(), ev_val.data());
            if (rc) return rc;
            bt.sweeps.assign(bt.n_sets, 0);
            bt.residual.assign(bt.n_sets, 0.0);
        }
        if ((rc = flush_batch_evidence(e))) return rc;
        rc = run_batch_launches(e, eps, max_sweeps);
        if (rc) return rc;
    }
    bt.have_run = true;
    e->stats.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    for (int32_t q = 0; q < bt.n_sets; ++q) {
        if (sweeps_out) sweeps_out[q] = bt.sweeps[q];
        if (residual_out) residual_out[q] = bt.residual[q];
    }
    return BN_OK;
}

extern "C" int bn_bp_copy_beliefs_batch(bn_engine* e, double* beliefs_out) {
    if (!e || !beliefs_out) return fail(BN_ERR_ARG, "null argument");
    if (e->batch_on_dense && e->dense) return bn_bp_copy_beliefs_batch(e->dense, beliefs_out);
    if (e->host_only || !e->batch.have_run) return fail(BN_ERR_STATE, "no batched run to copy from");
    if (e->batch.beliefs_on_host) {  // the last run wrote them into page-locked host memory
        std::memcpy(beliefs_out, e->batch.h_beliefs, sizeof(double) * size_t(e->batch.n_sets) * e->plan.node_off[e->plan.n]);
        return BN_OK;
    }
    ON_DEVICE(e);
    HIPCHK(hipMemcpyAsync(beliefs_out, e->batch.d_beliefs, sizeof(double) * size_t(e->batch.n_sets) * e->plan.node_off[e->plan.n],
                          hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return BN_OK;
}

extern "C" int bn_bp_residual_history_batch(bn_engine* e, int32_t set, double* out, int32_t cap) {
    if (!e || !out || cap < 0) return fail(BN_ERR_ARG, "bad argument");
    if (e->batch_on_dense && e->dense) return bn_bp_residual_history_batch(e->dense, set, out, cap);
    if (e->host_only || !e->batch.have_run) return fail(BN_ERR_STATE, "no batched run yet");
    if (set < 0 || set >= e->batch.n_sets) return fail(BN_ERR_ARG, "set index out of range");
    const int32_t cnt = std::min({cap, e->batch.sweeps[set], e->res_cap});
    ON_DEVICE(e);
    if (cnt > 0)
        HIPCHK(hipMemcpy(out, e->batch.d_res_hist + size_t(set) * e->res_cap, sizeof(double) * cnt, hipMemcpyDeviceToHost));
    return cnt;
}

extern "C" int bn_bp_run_batch(bn_engine* e, int32_t n_sets, const int32_t* ne, const int32_t* ev_node, const int32_t* ev_off,
                               const double* ev_val, double eps, int32_t max_sweeps, double* beliefs_out, int32_t* sweeps_out,
                               double* residual_out) {
    if (!beliefs_out) return fail(BN_ERR_ARG, "null beliefs_out");
    int rc = bn_bp_set_evidence_batch(e, n_sets, ne, ev_node, ev_off, ev_val);
    if (rc) return rc;
    bn_engine* on = (e->batch_on_dense && e->dense) ? e->dense : e;
    on->batch.direct_out = true;   // (the one-workgroup path writes the marginals into page-locked host memory; other paths ignore it)
    rc = bn_bp_run_batch_device(e, eps, max_sweeps, sweeps_out, residual_out);
    on->batch.direct_out = false;
    if (rc) return rc;
    return bn_bp_copy_beliefs_batch(e, beliefs_out);
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
        if (e->dag.uniform4) {
            HIPCHK(hipMemcpy(pi_msg_out, e->d_g_state + 2 * dag_off_pim(E, n, par, 0), sizeof(double) * 4 * size_t(E), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(lambda_msg_out, e->d_g_state + 2 * dag_off_lam(E, n, par, 0), sizeof(double) * 4 * size_t(E), hipMemcpyDeviceToHost));
            return BN_OK;
        }
        // arities below 4: the padded records, of which edge e's first k(parent of e) entries exist
        std::vector<double> pm(size_t(E) * 4), lm(size_t(E) * 4);
        HIPCHK(hipMemcpy(pm.data(), e->d_g_state + 2 * dag_off_pim(E, n, par, 0), sizeof(double) * 4 * size_t(E), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(lm.data(), e->d_g_state + 2 * dag_off_lam(E, n, par, 0), sizeof(double) * 4 * size_t(E), hipMemcpyDeviceToHost));
        size_t at = 0;
        for (int64_t ed = 0; ed < E; ++ed) {
            const int kp = e->plan.k[e->plan.in_idx[ed]];
            for (int i = 0; i < kp; ++i, ++at) { pi_msg_out[at] = pm[size_t(ed) * 4 + i]; lambda_msg_out[at] = lm[size_t(ed) * 4 + i]; }
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

// ---- layout introspection ---------------------------------------------------------------------
// The plan of the one-workgroup path (bn_small.hpp; tests emulate the kernel on it).  dims_out[12] = n, N, M, S, T, TT, CL,
// waves, re, rb, rc, mmax; the arrays (any may be null) are sized from those: ent [re * 64 waves][2], ent_cpt [re * 64 waves],
// term [TT], clist [CL], bslot / cslot [rb | rc * 64 waves][4], npi_init [N].  BN_ERR_STATE when the network is not eligible.
static int small_plan_copy(const SmallPlan& sp, const SmallPlan& tables, int32_t* dims_out, uint32_t* ent, double* ent_cpt, uint32_t* term,
                           uint16_t* clist, uint32_t* bslot, uint32_t* cslot, double* npi_init);
extern "C" int bn_small_plan_get(bn_engine* e, int32_t* dims_out, uint32_t* ent, double* ent_cpt, uint32_t* term, uint16_t* clist,
                                 uint32_t* bslot, uint32_t* cslot, double* npi_init) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    const SmallPlan& sp = e->small;
    if (!sp.ok) return fail(BN_ERR_STATE, "not eligible for the one-workgroup path: " + (sp.why.empty() ? std::string("disabled") : sp.why));
    return small_plan_copy(sp, sp, dims_out, ent, ent_cpt, term, clist, bslot, cslot, npi_init);
}
// ... of part `part` of the plan that spreads a mid-size network over several workgroups (bn_get_info "mid_parts"); the same
// layout (message / node-vector indices global, staging places the part's own); dims_out[12..13] = the part's node range
extern "C" int bn_mid_plan_get(bn_engine* e, int32_t part, int32_t* dims_out, uint32_t* ent, double* ent_cpt, uint32_t* term, uint16_t* clist,
                               uint32_t* bslot, uint32_t* cslot, double* npi_init) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (!e->mid.ok) return fail(BN_ERR_STATE, "not eligible for the mid-size path: " + (e->mid.why.empty() ? std::string("not needed or disabled") : e->mid.why));
    if (part < 0 || part >= int32_t(e->mid.parts.size())) return fail(BN_ERR_ARG, "part index out of range");
    const SmallPlan& sp = e->mid.parts[part];
    if (dims_out) { dims_out[12] = sp.v0; dims_out[13] = sp.v1; }
    return small_plan_copy(sp, e->mid.parts[0], dims_out, ent, ent_cpt, term, clist, bslot, cslot, npi_init);
}
static int small_plan_copy(const SmallPlan& sp, const SmallPlan& tables, int32_t* dims_out, uint32_t* ent, double* ent_cpt, uint32_t* term,
                           uint16_t* clist, uint32_t* bslot, uint32_t* cslot, double* npi_init) {
    if (dims_out) {
        const int32_t d[12] = {sp.n, sp.N, sp.M, sp.S, sp.T, sp.TT, sp.CL, sp.waves, sp.re, sp.rb, sp.rc, sp.mmax};
        std::copy(d, d + 12, dims_out);
    }
    if (ent) std::memcpy(ent, sp.ent.data(), sp.ent.size() * sizeof(SmallEntry));
    if (ent_cpt) std::copy(sp.ent_cpt.begin(), sp.ent_cpt.end(), ent_cpt);
    if (term) std::copy(sp.term.begin(), sp.term.begin() + sp.TT, term);
    if (clist) std::copy(sp.clist.begin(), sp.clist.begin() + sp.CL, clist);
    if (bslot) std::memcpy(bslot, sp.bslot.data(), sp.bslot.size() * sizeof(SmallSlot));
    if (cslot) std::memcpy(cslot, sp.cslot.data(), sp.cslot.size() * sizeof(SmallSlot));
    if (npi_init) std::copy(tables.npi_init.begin(), tables.npi_init.end(), npi_init);
    return BN_OK;
}

// The plan of the register-resident DAG path (bn_dag.hpp; tests emulate the kernel on it).
extern "C" int bn_dag_plan_get(bn_engine* e, int32_t* dims_out, int32_t* tiles, int32_t* slot_ptr, int32_t* cnode, int32_t* pitem,
                               int32_t* oedge, double* cpt_img, double* npi_init) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    const DagPlan& dp = e->dag;
    if (!dp.ok) return fail(BN_ERR_STATE, "not eligible for the register-resident DAG path: " + (dp.why.empty() ? std::string("disabled") : dp.why));
    if (dims_out) {
        const int32_t d[8] = {dp.n, dp.E, int32_t(dp.tiles.size()), dp.blocks, dp.stream ? 1 : 0, dp.n_child_tiles, dp.n_parent_tiles,
                              int32_t(dp.cpt_img.size())};
        std::copy(d, d + 8, dims_out);
    }
    if (tiles) std::memcpy(tiles, dp.tiles.data(), dp.tiles.size() * sizeof(DagTile));
    if (slot_ptr) std::copy(dp.slot_ptr.begin(), dp.slot_ptr.end(), slot_ptr);
    if (cnode) std::memcpy(cnode, dp.cnode.data(), dp.cnode.size() * sizeof(DagChildLane));
    if (pitem) std::memcpy(pitem, dp.pitem.data(), dp.pitem.size() * sizeof(DagParentLane));
    if (oedge) std::copy(dp.oedge.begin(), dp.oedge.end(), oedge);
    if (cpt_img) std::copy(dp.cpt_img.begin(), dp.cpt_img.end(), cpt_img);
    if (npi_init) std::copy(dp.npi_init.begin(), dp.npi_init.end(), npi_init);
    return BN_OK;
}

// New CPT values on an unchanged structure (belief_propagation.hpp:61,186,252: the reference reads node->cpt on every call, so a
// table edited or re-fitted after the functor was built IS seen there; here the tables are device images made at bn_create).
// Re-derives every image that holds CPT values -- the lane-striped tile image, the entry tables of the item kernels, the
// register image of the DAG path, the initial pi(v) of the roots -- from the new flat array and copies them over the old ones;
// the sampler state is rebuilt at its next call.  No allocation changes size.
template <class T>
static int reupload(T* dst, const std::vector<T>& src, size_t expect, hipStream_t s, const char* what) {
    if (src.size() != expect) return fail(BN_ERR_STATE, std::string("bn_reload_cpt: the ") + what + " changed size (structure changed?)");
    if (!src.empty()) HIPCHK(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return BN_OK;
}
extern "C" int bn_reload_cpt(bn_engine* e, const double* cpt, int64_t n_entries) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: an earlier bn_reload_cpt failed while uploading (destroy it and create a new one)");
    Plan& p = e->plan;
    const int64_t want = p.n > 0 ? p.cpt_off[p.n] : 0;
    if (n_entries != want) return fail(BN_ERR_ARG, "bn_reload_cpt: " + std::to_string(n_entries) + " entries given, the model has " + std::to_string(want));
    if (want > 0 && !cpt) return fail(BN_ERR_ARG, "null cpt");
    // Two phases.  (1) every host plan that holds CPT values is rebuilt from the new array into TEMPORARIES and checked against
    // the one in use; a mismatch returns with the engine exactly as it was.  (2) the temporaries are swapped in and the device
    // images overwritten; a HIP failure there leaves device images of mixed age, so the engine is marked unusable.
    std::vector<double> old_flat;
    bool swapped_flat = false;
    try {
        old_flat.assign(cpt, cpt + want);
        old_flat.swap(p.cpt_flat);   // the planners read p.cpt_flat; old_flat now holds the values in force
        swapped_flat = true;
        auto refuse = [&](const char* what) {
            p.cpt_flat.swap(old_flat);
            return fail(BN_ERR_STATE, std::string("bn_reload_cpt: ") + what);
        };
        SmallPlan n_small;
        MidPlan n_mid;
        DagPlan n_dag;
        if (e->small.ok) {
            build_small_plan(p, n_small);
            if (!n_small.ok || n_small.ent_cpt.size() != e->small.ent_cpt.size() || n_small.npi_init.size() != e->small.npi_init.size())
                return refuse("the one-workgroup plan changed");
        }
        if (e->mid.ok) {
            build_mid_plan(p, n_mid);
            bool same = n_mid.ok && n_mid.parts.size() == e->mid.parts.size();
            for (size_t q = 0; same && q < n_mid.parts.size(); ++q) same = n_mid.parts[q].ent_cpt.size() == e->mid.parts[q].ent_cpt.size();
            same = same && (n_mid.parts.empty() || n_mid.parts[0].npi_init.size() == e->mid.parts[0].npi_init.size());
            if (!same) return refuse("the plan of the several-workgroup path changed");
        }
        if (e->dag.ok) {
            build_dag_plan(p, e->host_only ? 224 : std::max(e->dag.blocks, 8), n_dag);   // (the same cap gives the same plan; only the values differ)
            if (!n_dag.ok || n_dag.cpt_img.size() != e->dag.cpt_img.size() || n_dag.blocks != e->dag.blocks || n_dag.npi_init.size() != e->dag.npi_init.size())
                return refuse("the plan of the register-resident DAG path changed");
        }
        if (!e->host_only) {
            ON_DEVICE(e);
            if (hipStreamSynchronize(e->stream) != hipSuccess) return refuse("the engine's stream reports an error");   // nothing of the old tables is in use any more
            stripe_cpt(p, cpt);
            if (p.cpt_striped.size() != size_t(p.cpt_doubles)) { std::vector<double>().swap(p.cpt_striped); return refuse("the tile image changed size"); }
        }
        // ---- commit
        if (e->small.ok) e->small = std::move(n_small);
        if (e->mid.ok) e->mid = std::move(n_mid);
        if (e->dag.ok) e->dag = std::move(n_dag);
        if (e->dense) { free_engine(e->dense); e->dense = nullptr; e->batch_on_dense = false; }   // (rebuilt from the new tables on demand)
        if (e->host_only) return BN_OK;
        e->poisoned = true;   // until every image has arrived
        ON_DEVICE(e);
        hipStream_t s = e->stream;
        int rc;
        if ((rc = reupload(e->d_cpt, p.cpt_striped, size_t(p.cpt_doubles), s, "tile image"))) return rc;
        if (e->small_ok) {
            if ((rc = reupload(e->d_s_cpt, e->small.ent_cpt, e->small.ent_cpt.size(), s, "entry table"))) return rc;
            if ((rc = reupload(e->d_s_init, e->small.npi_init, e->small.npi_init.size(), s, "initial pi"))) return rc;
        }
        if (e->mid_ok) {
            std::vector<double> all;
            for (const SmallPlan& sp : e->mid.parts) all.insert(all.end(), sp.ent_cpt.begin(), sp.ent_cpt.end());
            if ((rc = reupload(e->d_m_cpt, all, all.size(), s, "entry tables"))) return rc;
            HIPCHK(hipStreamSynchronize(s));   // `all` is a local
            if ((rc = reupload(e->d_m_init, e->mid.parts[0].npi_init, e->mid.parts[0].npi_init.size(), s, "initial pi"))) return rc;
        }
        if (e->dag_ok) {
            if ((rc = reupload(e->d_g_cpt, e->dag.cpt_img, e->dag.cpt_img.size(), s, "register image"))) return rc;
            if ((rc = reupload(e->d_g_init, e->dag.npi_init, e->dag.npi_init.size(), s, "initial pi"))) return rc;
        }
        HIPCHK(hipStreamSynchronize(s));
        std::vector<double>().swap(p.cpt_striped);
        lw_free(e->lw);   // the sampler uploads its copy of the tables at its next call
        e->poisoned = false;
    } catch (const std::bad_alloc&) {
        if (swapped_flat && !e->poisoned) p.cpt_flat.swap(old_flat);   // phase 1: nothing was committed
        return fail(BN_ERR_ALLOC, "out of host memory in bn_reload_cpt");
    }
    return BN_OK;
}

extern "C" int bn_layout_get(bn_engine* e, bn_layout_info* o) {
    if (!e || !o) return fail(BN_ERR_ARG, "null argument");
    const Plan& p = e->plan;
    o->n_nodes = p.n;
    o->n_edges = int32_t(p.E);
    o->n_classes = int32_t(p.classes.size());
    o->n_tiles = int32_t(p.tiles.size());
    o->lanes_per_node_max = p.g_max;
    o->cpt_doubles = p.cpt_doubles;
    o->rec_doubles = p.rec_doubles;
    o->node_doubles = p.node_doubles;
    o->algorithmic_bytes_per_sweep = p.algorithmic_bytes;
    o->layout_bytes_per_sweep = p.layout_bytes;
    o->messages_per_sweep = p.messages_per_sweep;
    o->rank = p.rank;
    o->nranks = p.nranks;
    o->n_owned = p.n_owned;
    o->n_interior_tiles = p.n_interior_tiles;
    o->n_cut_edges = p.n_cut_edges;
    o->segment_bytes = p.seg_d2 * 16;
    o->segment_used_bytes = p.seg_used_d2.empty() ? 0 : p.seg_used_d2[p.rank] * 16;
    o->exchange_base = p.g_base;
    return BN_OK;
}

extern "C" int bn_layout_node_tiles(bn_engine* e, int32_t* tiles_out) {
    if (!e || !tiles_out) return fail(BN_ERR_ARG, "null argument");
    std::copy(e->plan.node_tile.begin(), e->plan.node_tile.end(), tiles_out);
    return BN_OK;
}

extern "C" int bn_layout_node_slots(bn_engine* e, int32_t* slots_out) {
    if (!e || !slots_out) return fail(BN_ERR_ARG, "null argument");
    std::copy(e->plan.node_slot.begin(), e->plan.node_slot.end(), slots_out);
    return BN_OK;
}

extern "C" int bn_layout_edge_refs(bn_engine* e, int32_t* pi_out, int32_t* lam_out) {
    if (!e || !pi_out || !lam_out) return fail(BN_ERR_ARG, "null argument");
    for (int64_t i = 0; i < e->plan.E; ++i) { pi_out[i] = e->plan.edge_ref[i].pi; lam_out[i] = e->plan.edge_ref[i].lam; }
    return BN_OK;
}

extern "C" int bn_layout_class(bn_engine* e, int32_t cls, int32_t* kv, int32_t* m, int32_t* lanes_per_node,
                               int32_t* variant, int32_t* n_nodes) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (cls < 0 || cls >= int32_t(e->plan.classes.size())) return fail(BN_ERR_ARG, "class index out of range");
    const ClassDesc& c = e->plan.classes[cls];
    if (kv) *kv = c.kv;
    if (m) *m = c.m;
    if (lanes_per_node) *lanes_per_node = c.G;
    if (variant) *variant = c.variant;
    if (n_nodes) *n_nodes = c.n_nodes;
    return BN_OK;
}

// ---- likelihood weighting -------------------------------------------------------------------------
extern "C" int bn_lw_run(bn_engine* e, int32_t ne, const int32_t* ev_node, const int32_t* ev_state,
                         uint64_t sample_begin, uint64_t n_samples, uint64_t seed, double* hist_out) {
    if (!e || !hist_out) return fail(BN_ERR_ARG, "null argument");
    if (e->host_only) return fail(BN_ERR_STATE, "engine was created with BN_DEVICE_HOST_ONLY: no GPU, no compute");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: bn_reload_cpt failed while uploading (destroy it and create a new one)");
    if (ne < 0 || (ne > 0 && (!ev_node || !ev_state))) return fail(BN_ERR_ARG, "bad evidence arguments");
    ON_DEVICE(e);
    std::string err;
    int rc = lw_run(e->lw, e->plan, e->stream, ne, ev_node, ev_state, sample_begin, n_samples, seed, hist_out, err);
    if (rc) return fail(rc, err);
    return BN_OK;
}

// Likelihood weighting over every rank of the communicator: the sample range is split evenly,
// each GPU draws its share (disjoint sample ids = disjoint streams), ONE RCCL all-reduce sums the histograms.
extern "C" int bn_lw_run_allreduce(bn_engine* e, int32_t ne, const int32_t* ev_node, const int32_t* ev_state,
                                   uint64_t sample_begin, uint64_t n_samples_total, uint64_t seed, double* hist_out) {
    if (!e || !hist_out) return fail(BN_ERR_ARG, "null argument");
    if (e->host_only) return fail(BN_ERR_STATE, "engine was created with BN_DEVICE_HOST_ONLY: no GPU, no compute");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: bn_reload_cpt failed while uploading (destroy it and create a new one)");
    if (ne < 0 || (ne > 0 && (!ev_node || !ev_state))) return fail(BN_ERR_ARG, "bad evidence arguments");
    if (!e->comm) return fail(BN_ERR_COMM, "call bn_comm_init first");
    ON_DEVICE(e);
    const uint64_t P = uint64_t(e->plan.nranks), r = uint64_t(e->plan.rank);
    const uint64_t lo = n_samples_total * r / P, hi = n_samples_total * (r + 1) / P;
    std::string err;
    int rc = lw_run(e->lw, e->plan, e->stream, ne, ev_node, ev_state, sample_begin + lo, hi - lo, seed, nullptr, err);
    if (rc) return fail(rc, err);
    const size_t hist_n = size_t(e->plan.node_off[e->plan.n]);
    ncclResult_t nr = g_rccl.AllReduce(e->lw.d_hist, e->lw.d_hist, hist_n, ncclDouble, ncclSum, e->comm, e->stream);
    if (nr != ncclSuccess) return fail(BN_ERR_COMM, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(nr));
    HIPCHK(hipMemcpyAsync(hist_out, e->lw.d_hist, hist_n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return BN_OK;
}

// Rejection (logic) sampling, reference rejection_sampling.hpp:33-167.
extern "C" int bn_rs_run(bn_engine* e, int32_t ne, const int32_t* ev_node, const int32_t* ev_state,
                         uint64_t sample_begin, uint64_t n_accept, uint64_t max_draw, uint64_t seed, double* counts_out,
                         uint64_t* drawn_out, uint64_t* accepted_out) {
    if (!e || !counts_out) return fail(BN_ERR_ARG, "null argument");
    if (e->host_only) return fail(BN_ERR_STATE, "engine was created with BN_DEVICE_HOST_ONLY: no GPU, no compute");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: bn_reload_cpt failed while uploading (destroy it and create a new one)");
    if (ne < 0 || (ne > 0 && (!ev_node || !ev_state))) return fail(BN_ERR_ARG, "bad condition arguments");
    if (max_draw == 0) return fail(BN_ERR_ARG, "max_draw must be > 0 (the reference loops forever on impossible evidence)");
    ON_DEVICE(e);
    std::string err;
    int rc = rs_run(e->lw, e->plan, e->stream, ne, ev_node, ev_state, sample_begin, n_accept, max_draw, seed,
                    counts_out, drawn_out, accepted_out, err);
    if (rc) return fail(rc, err);
    return BN_OK;
}

// CPT fitting from a table of joint patterns, reference sampler.hpp:81-163 (sampler::make_cpt).
// Stateless: only the structure of `desc` is read (its cpt pointer is ignored and may be null).
extern "C" int bn_fit_cpt(const bn_model_desc* desc, int64_t n_patterns, const uint8_t* patterns,
                          const uint64_t* counts, double* cpt_out) {
    if (!desc || !cpt_out) return fail(BN_ERR_ARG, "null argument");
    if (desc->n_nodes < 0 || (desc->n_nodes > 0 && (!desc->k || !desc->in_ptr || !desc->cpt_off)))
        return fail(BN_ERR_ARG, "bad model structure");
    if (n_patterns < 0 || (n_patterns > 0 && (!patterns || !counts))) return fail(BN_ERR_ARG, "bad pattern table");
    const int32_t n = desc->n_nodes;
    std::vector<int32_t> k(desc->k, desc->k + n), in_ptr(desc->in_ptr, desc->in_ptr + n + 1);
    std::vector<int64_t> cpt_off(desc->cpt_off, desc->cpt_off + n + 1);
    if (in_ptr[0] != 0 || cpt_off[0] != 0) return fail(BN_ERR_ARG, "in_ptr / cpt_off must start at 0");
    for (int32_t v = 0; v < n; ++v) {
        if (k[v] < 1 || k[v] > 255) return fail(BN_ERR_ARG, "node arity must be in 1..255");
        if (in_ptr[v + 1] < in_ptr[v] || in_ptr[v + 1] - in_ptr[v] > BN_MAX_PARENTS) return fail(BN_ERR_ARG, "bad in_ptr");
    }
    if (in_ptr[n] > 0 && !desc->in_idx) return fail(BN_ERR_ARG, "bad model structure");
    std::vector<int32_t> in_idx(desc->in_idx, desc->in_idx + in_ptr[n]);
    for (int32_t v = 0; v < n; ++v) {
        int64_t rows = 1;
        for (int32_t e = in_ptr[v]; e < in_ptr[v + 1]; ++e) {
            if (in_idx[e] < 0 || in_idx[e] >= n || in_idx[e] == v) return fail(BN_ERR_ARG, "parent index out of range");
            rows *= k[in_idx[e]];
        }
        if (cpt_off[v + 1] - cpt_off[v] != rows * k[v]) return fail(BN_ERR_ARG, "cpt_off does not match the arities");
    }
    // sampler::make_cpt returns false on an empty table (:83); here that is an argument error
    uint64_t total = 0;
    for (int64_t i = 0; i < n_patterns; ++i) total += counts[i];
    if (total == 0) return fail(BN_ERR_ARG, "empty sample table (sampling_size() == 0)");
    // [node][pattern] image so that a wave reads contiguous bytes
    std::vector<uint8_t> tr(std::max<size_t>(size_t(n) * size_t(n_patterns), 1));
    for (int64_t i = 0; i < n_patterns; ++i)
        for (int32_t v = 0; v < n; ++v) {
            if (patterns[i * n + v] >= k[v]) return fail(BN_ERR_ARG, "pattern state out of range");
            tr[size_t(v) * n_patterns + i] = patterns[i * n + v];
        }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(BN_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (desc->device >= ndev) return fail(BN_ERR_ARG, "device ordinal out of range");
    if (desc->device >= 0) HIPCHK(hipSetDevice(desc->device));
    std::vector<int32_t> row_node;
    std::vector<int64_t> row_off;
    for (int32_t v = 0; v < n; ++v)
        for (int64_t o = cpt_off[v]; o < cpt_off[v + 1]; o += k[v]) { row_node.push_back(v); row_off.push_back(o); }
    const size_t entries = size_t(cpt_off[n]);
    hipStream_t s = nullptr;
    uint8_t* d_pat = nullptr; unsigned long long* d_w = nullptr; unsigned long long* d_cnt = nullptr;
    int32_t *d_k = nullptr, *d_ptr = nullptr, *d_idx = nullptr, *d_rn = nullptr;
    int64_t *d_off = nullptr, *d_ro = nullptr; double* d_out = nullptr;
    int rc = [&]() -> int {
        int r;
        HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        if ((r = upload(&d_pat, tr, s))) return r;
        std::vector<unsigned long long> w(counts, counts + n_patterns);
        if ((r = upload(&d_w, w, s))) return r;
        if ((r = upload(&d_k, k, s))) return r;
        if ((r = upload(&d_ptr, in_ptr, s))) return r;
        if ((r = upload(&d_idx, in_idx, s))) return r;
        if ((r = upload(&d_off, cpt_off, s))) return r;
        if ((r = upload(&d_rn, row_node, s))) return r;
        if ((r = upload(&d_ro, row_off, s))) return r;
        if ((r = dalloc(&d_cnt, entries))) return r;
        if ((r = dalloc(&d_out, entries))) return r;
        HIPCHK(hipMemsetAsync(d_cnt, 0, std::max<size_t>(entries, 1) * 8, s));
        FitArgs a{n, d_k, d_ptr, d_idx, d_off, n_patterns, d_pat, d_w, d_cnt, int64_t(row_node.size()), d_rn, d_ro, d_out};
        if (launch_fit(a, s)) return fail(BN_ERR_HIP, "fit kernel launch failed");
        if (entries) HIPCHK(hipMemcpyAsync(cpt_out, d_out, entries * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return BN_OK;
    }();
    void* ptrs[] = {d_pat, d_w, d_cnt, d_k, d_ptr, d_idx, d_rn, d_off, d_ro, d_out};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    if (s) (void)hipStreamDestroy(s);
    return rc;
}

extern "C" int bn_lw_states(bn_engine* e, uint64_t n, uint8_t* states_out, double* weights_out) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (e->host_only) return fail(BN_ERR_STATE, "host-only engine");
    ON_DEVICE(e);
    std::string err;
    int rc = lw_states(e->lw, e->plan, e->stream, n, states_out, weights_out, err);
    if (rc) return fail(rc, err);
    return BN_OK;
}

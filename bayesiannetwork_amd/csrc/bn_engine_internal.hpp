// bn_engine_internal.hpp -- what the translation units of the C ABI (include/bn_mi355x.h) share: the engine object and the helpers
// that more than one of them uses.  bn_engine.cpp: creation, evidence, the run of a single query (steps, the one-launch paths and
// their dispatch), options, read-out; bn_engine_batch.cpp: several evidence sets per call; bn_engine_shard.cpp: RCCL communicator
// and the in-kernel exchange of sharded engines; bn_engine_tools.cpp: plan / layout introspection, bn_reload_cpt, the samplers' and
// the fit's entry points.  No CPU compute path exists in any of them: every result comes from the kernels.
#ifndef BN_ENGINE_INTERNAL_HPP
#define BN_ENGINE_INTERNAL_HPP
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

struct bn_engine;
namespace bn_eng __attribute__((visibility("hidden"))) {
extern thread_local std::string g_err;   // bn_last_error (per thread)
int fail(int code, const std::string& msg);
}  // namespace bn_eng
using namespace bn_eng;


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
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;   // (optional: bn_get_info "rccl_ranks")
};

struct bn_engine {
    Plan plan;
    bool host_only = true;
    int64_t create_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // bn_get_info "create_us_{plan,small,mid,dag,device}"
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
        bool dag_ev_applied = false;  // state slot q holds set q's evidence under mark dag_mark (a batch of at most kDagMaxSets sets on a network
                                      // without padding: an observed node's vectors are carried over by every sweep, so they outlive the run)
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
    bool dense_refused = false;     // the dense layout would give some node's sums another order than this engine's single queries: batches stay here
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
    bool dag_ok = false;            // eligible on this device (the LIGHT plan is in e->dag)
    bool dag_flow_ok = false;       // the plan has a dataflow form (<= 64 neighbour tiles per tile, one tile per wave, > 1 block) and the service block fits
    int dag_flow = 0;               // option "dagflow" 1: single queries take the dataflow form where it exists (default 0: measured slower, EXPERIMENTS R6.2)
    int32_t dag_flow_pause = 0;     // runs left on the barrier form after a dataflow launch gave up a wait
    int32_t dag_flow_max_nbr = 0;
    int last_dag_flow = 0;
    int32_t* d_g_nbr = nullptr;
    DagFlowSync* d_g_flow = nullptr;
    bool dag_ready = false;         // full plan built, device tables and image uploaded (ensure_dag)
    int32_t dag_cap = 224;          // the block cap the plan was built for
    int dag_mode = 1;               // option "dag": 0 never, 1 where eligible and no other one-launch path takes the network, 2 wherever eligible
    int32_t dag_cooldown = 0, dag_aborts = 0;   // runs left on the tile kernels after a grid wait gave up; how often that happened
    DagTile* d_g_tiles = nullptr;
    int32_t* d_g_slotptr = nullptr;
    DagChildLane* d_g_cnode = nullptr;      // (tiles, parent lanes and out-edges: with state records in place of CSR ids, build_dag_device_tables)
    DagParentLaneDev* d_g_pitem = nullptr;
    int32_t* d_g_oedge = nullptr;
    int32_t* d_g_eperm = nullptr;           // CSR edge id -> message record, node id -> slot of its vectors
    int32_t* d_g_nperm = nullptr;
    DagDeviceTables dag_tables;             // (host copy, built with the plan -- also on host-only engines, where the CPU sanitizer run walks it; bn_bp_messages reads eperm)
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

namespace bn_eng __attribute__((visibility("hidden"))) {
extern RcclApi g_rccl;
int load_rccl();

template <class T, class A>
inline int upload(T** dst, const std::vector<T, A>& src, hipStream_t s) {
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIPCHK(hipMalloc(reinterpret_cast<void**>(dst), bytes));
    if (!src.empty()) HIPCHK(hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return BN_OK;
}

template <class T>
inline int dalloc(T** dst, size_t count) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(dst), std::max<size_t>(count, 1) * sizeof(T)));
    return BN_OK;
}


bool dag_applies(const bn_engine* e);
int ensure_dag(bn_engine* e);
SmallArgs small_args_of(bn_engine* e, const BpBuffers& b, double eps, int32_t max_sweeps, int32_t begin, Ctl* host_ctl);
int mid_launch(bn_engine* e, const MidArgs& a, int32_t n_sets, const double* copy_from, double* copy_to, bool wait = true);
MidArgs mid_args_of(bn_engine* e, const BpBuffers& b0, const SetStrides& st, Ctl* h_ctl_dev, double eps, int32_t max_sweeps,
                           int32_t begin, int32_t set_base, int32_t slot_base);
bool mid_applies(const bn_engine* e);
BpBuffers buffers_of(bn_engine* e);
int run_dag(bn_engine* e, double eps, int32_t max_sweeps, double* copy_to);
int resident_service_blocks(int tile_blocks);
void report_abort_once(bn_engine* e, const char* what, int pause_runs);
int mid_reserve_slots(bn_engine* e, int32_t slots);
int ensure_events(bn_engine* e, size_t count);
int check_evidence(const Plan& p, int32_t ne, const int32_t* ev_node, const int32_t* ev_off,
                          std::vector<uint32_t>& seen, uint32_t& epoch);
void free_engine(bn_engine* e);
int small_gave_up(bn_engine*);
void resident_ran_ok(bn_engine* e);
}  // namespace bn_eng

#endif  // BN_ENGINE_INTERNAL_HPP

// bn_dag.hpp -- networks of arity <= 4 (2 and 3 padded to 4: bn_dag_plan.cpp) with up to 5 parents per node (BASELINE.json configs[1]: the 10 k-node random DAG, 2.7 M
// CPT entries): the whole run in ONE launch with every CPT entry resident in a register (bn_dag.hip).
//
// The tile layout (bn_plan.hpp) runs such a network at 12.4 us per sweep: 2.9 us of launch gap, then ONE lane-group tile's
// latency -- descriptor -> CPT stream -> references -> records -> 1 280 dependent fp64 instructions -> parent role, in a
// row on one wave.  The item kernels (bn_small.hpp) stage 5 terms per entry in LDS: 100 MB here, the chip has 40.
// This path splits a node's two roles (belief_propagation.hpp) over DIFFERENT waves and keeps everything static on chip:
//   child tile   one wavefront of nodes with m parents (64 nodes, or 64 / 4^(m-2) lane groups for m >= 3): pi(v) =
//                calculate_pi (:174-200) and the lambda-message to every parent = calculate_lambda_k (:240-266).  Its 64
//                CPT entries per lane live in VGPRs for the whole run (2.7 M entries = 17 % of the chip's register files).
//                m <= 2: the reference's operation order, bit for bit.  m >= 3: the contraction is FACTORED -- the sums over
//                the two trailing parents are formed once and shared by all five outputs (350 instead of 1 280 fp64
//                operations per lane) and combined across the group's lanes by shuffles; results agree with the reference
//                to rounding (its own >= 3-parent products are unordered: it iterates an unordered_map, :253).
//   parent item  one lane per pi-message (calculate_pi_i, :202-218) and one per lambda(v) (calculate_lambda, :220-238):
//                the product over the node's children in ascending order, skipping the target -- the reference's order.
// State (pi-/lambda-messages in CSR edge order, pi(v), lambda(v); double-buffered) lives in device memory and is exchanged
// with 16-byte sc1 accesses; one grid barrier per iteration in the atomic-free granule form of bn_resident.hip (every block
// publishes {generation | residual half} pairs, the first wave of every block collects them: one hand-off per iteration).
// Networks beyond what the chip holds at one tile per wave run the same tile code in STREAM form: a wave walks several
// tiles per iteration and re-reads their CPT image from L2 / memory.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "bn_device.hpp"

namespace bnmi {

constexpr int kDagWaves = 8;            // per block: two per SIMD, 256 VGPRs each
constexpr int kDagMaxBlocks = kResidentMaxBlocks;  // one granule pair per block, four pairs per polling lane
constexpr int kDagBudget = kResidentBudget;        // iterations per launch (ResidentSync::res)
constexpr int kDagMaxParents = 5;
constexpr int kDagMaxChildren = 1024;   // per node: DagParentLane::deg_tpos holds count | rank << 16 (signed), wide items cost deg^2 loads
constexpr int64_t kDagMaxImageBytes = int64_t(256) << 20;   // padded CPT image (registers, or re-read per sweep in stream form)
constexpr int kDagRegChildren = 8;
constexpr int kDagMaxSets = 16;     // evidence sets one launch can walk (bn_bp_run_batch): per-set state, marks, barrier words; the CPT registers serve all      // out-edge ids a parent item keeps in registers (more: re-read every iteration)

enum : int32_t { kDagChild0 = 0, /* 1..5: child tile of nodes with that many parents */ kDagParent = 8, kDagParentWide = 9 };

// One wavefront of work.  32 bytes.
struct DagTile {
    int32_t kind;       // 0..5: child tile, nodes with `kind` parents; kDagParent: parent items, a node's in adjacent lanes of this wave;
                        // kDagParentWide: items of nodes with more than 63 children
    int32_t n_active;   // nodes (child tile) / items (parent tile) in use
    int32_t lane_base;  // first entry of the tile in the per-lane tables (cnode / pitem): 64 per tile
    int32_t cpt_base;   // child tile: first double2 of its CPT image; entry pair q of lane l at cpt_base + q * 64 + l
    int32_t dmax;       // parent tile: largest child count among its items
    int32_t rec_base;   // child tile, device copy only (build_dag_device_tables): state record of in-edge 0 of its first node -- record j of
    int32_t slot_base;  // its i-th node at rec_base + j * n_active + i -- and the slot of that node's vectors (the i-th node's: slot_base + i)
    int32_t pad_;
};
static_assert(sizeof(DagTile) == 32, "DagTile is loaded as two 16-byte words");

struct DagChildLane { int32_t node, ebase; };                 // node id (-1: idle lane), CSR id of its first in-edge
struct DagParentLane { int32_t node, tedge, obeg, deg_tpos; }; // node (-1: idle), CSR id of the target out-edge (-1: the lambda(v)
                                                              // item), first entry in oedge, child count | target's rank << 16

// What the kernels read instead of the tile and parent-lane tables (built at upload, build_dag_device_tables): the same entries with
// STATE RECORD numbers in place of CSR ids (a child tile's lanes need none of their own: tile bases + the lane's node index).  Message records and node-vector slots are numbered tile-major -- record j of the i-th node of a child
// tile at tile_base + j * (nodes in the tile) + i -- so that a load instruction of a child tile (lane i, record j) touches consecutive
// records instead of every M-th (CSR order: 64 lanes x 16 bytes out of 64 lanes x 32 M bytes; a CU's eight waves share one address
// pipe, and the tiles with the heaviest arithmetic are the ones that ask for most records).  The plan, its getters and the CPU
// emulator keep CSR ids; eperm / nperm (CSR edge id -> message record, node id -> vector slot) translate where the state is read
// or written by id: evidence, the padded networks' initial state, bn_bp_messages.
struct DagParentLaneDev { int32_t node, tedge, obeg, deg_tpos, snode, pad_[3]; };   // as DagParentLane, tedge = a record; slot of the node's vectors
static_assert(sizeof(DagParentLaneDev) == 32, "loaded as 16-byte words");

struct DagPlan {
    bool ok = false;
    bool light = false;                    // built without the CPT image (build_dag_plan's `light`)
    std::string why;
    int32_t n = 0, E = 0;
    std::vector<DagTile> tiles;            // in slot order: the tiles of wave slot s are [slot_ptr[s], slot_ptr[s + 1])
    std::vector<int32_t> slot_ptr;         // [blocks * kDagWaves + 1]; slot = block * kDagWaves + wave
    int32_t blocks = 0;
    bool stream = false;                   // some wave walks more than one tile per iteration
    bool has_groups = false;               // some node has 3..5 parents (lane-group tiles)
    double fill = 1.0;                     // real CPT entries / entries of the padded tables
    bool uniform4 = true;                  // every arity is 4; false: arities 2..4 padded to 4, the initial state is written to memory before a run
    std::vector<DagChildLane> cnode;       // [n_tiles * 64] (child tiles' entries)
    std::vector<DagParentLane> pitem;      // [n_tiles * 64] (parent tiles' entries)
    std::vector<int32_t> oedge;            // out-edges (CSR edge ids) of every node, children ascending
    BigVec cpt_img;                        // child tiles' images, double2 units x 2
    std::vector<double> npi_init;          // [n][4] initial pi(v): the CPT row of a root (:58-64, not normalised), else 1.0; 0 beyond the node's arity
    int32_t n_child_tiles = 0, n_parent_tiles = 0;
};

// cap_blocks: most blocks a launch may have (0.9 x CUs, rounded down to a multiple of 8; host-only engines: 224)
// light: everything but the CPT image (cpt_img stays empty) -- what the default-path policy and the layout getters need; the engine
// fills the image, builds the device tables and uploads them when the path is first wanted (bn_eng::ensure_dag)
void build_dag_plan(const Plan& p, int32_t cap_blocks, DagPlan& dp, bool light = false);
struct DagDeviceTables {
    std::vector<DagTile> tiles;
    std::vector<DagParentLaneDev> pitem;
    std::vector<int32_t> oedge, eperm, nperm;
};
void build_dag_device_tables(const DagPlan& dp, DagDeviceTables& dt);

// ---- dataflow form of the single query (bn_dag.hip "flow"): no grid barrier.  Every tile (= wave) publishes a granule pair
// {generation | residual half} per iteration; a tile starts its next iteration once its neighbour tiles (DagFlowTables::nbr) carry
// the generation of the previous one; one more block -- the service block -- collects all tiles' granules of iteration i while the
// tiles compute i + 1 and publishes {generation, verdict}; a tile starts i + 2 only once the verdict of i is known, so the state the
// run stops in is intact when it does (the one speculative iteration writes the OTHER buffer).  bn_resident.hip's protocol.
struct DagFlowTables {
    bool ok = false;
    int32_t max_nbr = 0;
    std::vector<int32_t> nbr;   // [tiles][kWave] tile indices (the plan's tile order), -1 = none
};
void build_dag_flow_tables(const DagPlan& dp, const Plan& p, DagFlowTables& ft);
enum : unsigned { kDagFlowGoOn = 0, kDagFlowConverged = 1, kDagFlowCapped = 2, kDagFlowAbort = 3, kDagFlowBudget = 4 };
struct DagFlowSync {            // zeroed at creation, after an aborted launch and before the generation would wrap
    struct Line {
        unsigned long long word;        // low 32 bits: generation of the last decided iteration; high 32: its verdict (kDagFlow*)
        unsigned long long pad_[15];
    } verdict[8];                       // copies on lines of their own: tile blocks poll copy blockIdx % 8
    unsigned long long res[kDagBudget]; // per-iteration maximum_difference, bit patterns
    // then: granule pairs [2 iteration parities][tiles][2]
};
inline size_t dag_flow_sync_bytes(size_t tiles) { return sizeof(DagFlowSync) + sizeof(unsigned long long) * 2 * tiles * 2; }

struct DagArgs {
    BpBuffers b;              // beliefs, res_hist / res_cap
    double eps;
    int32_t max_sweeps, sweep_begin, budget;
    uint32_t run_id, gen_base;
    unsigned long long timeout_ticks;
    ResidentSync* sync;       // abort word, granule tables, per-iteration residuals
    Ctl* host_ctl;
    unsigned* host_abort;
    int32_t n, E, n_blocks;
    const DagTile* tiles;
    const int32_t* slot_ptr;
    const DagChildLane* cnode;
    const DagParentLaneDev* pitem;
    const int32_t* oedge;
    const double* cpt_img;
    const double* npi_init;
    double* state;            // pi-messages [2][E][4], lambda-messages [2][E][4], pi(v) [2][n][4], lambda(v) [2][n][4]
    const uint8_t* frz;       // [n] evidence marks: observed when == frz_mark
    uint8_t frz_mark;
    int32_t poll_sleep;       // pause between two polls of the barrier, x 64 cycles
    int32_t first_poll_delay; // ... and between a block's arrival and its first poll
    // several evidence sets per launch (n_sets > 1; a single query: n_sets = 1, strides unused): set q's state at state + q * state_stride
    // (doubles), its marks at frz + q * frz_stride, its barrier words in sync[q], its control block host_ctl[q], its marginals at
    // b.beliefs + q * belief_stride, its residual history at b.res_hist + q * res_hist_stride
    int32_t n_sets;
    uint32_t set_mask;
    int64_t state_stride, frz_stride, belief_stride;
    int32_t res_hist_stride;
    // arities below 4 (padded to 4, DagPlan::uniform4 == false): the run's initial state stands in memory (dag_init_kernel), nothing
    // is synthesised in sweep 0; node_k / node_off say which entries of a node's padded vectors exist and where its marginal goes
    int32_t state_init;
    const int32_t* node_k;     // nullptr: every arity is 4
    const int64_t* node_off;
    // dataflow form (flow != nullptr: a single query on a plan with DagFlowTables::ok; the launch has n_blocks + 1 blocks)
    DagFlowSync* flow;
    const int32_t* nbr;        // [tiles][kWave]
    int32_t n_tiles;
    int32_t flow_sleep;        // pause between two polls of a tile's neighbours, x 8 x 64 cycles
};
struct DagEvidenceArgs {
    int32_t ne, n, E;
    const int32_t* ev_node;
    const int32_t* ev_off;
    const double* ev_val;
    double* state;
    uint8_t* frz;
    uint8_t frz_mark;
    const int32_t* node_k;     // nullptr: every arity is 4
    const int32_t* nperm;      // node id -> slot of its vectors in the state
};
// the initial state of a run of a padded network, buffer parity 0: messages = ones over the PARENT's states (:38-56), pi(v) = npi_init,
// lambda(v) = ones over the node's states, zeros in the padding; nodes that carry the evidence mark keep their vectors
struct DagInitArgs {
    int32_t n, E;
    const int32_t* in_ptr;     // [n + 1] CSR of the in-edges (edge e belongs to child c: in_ptr[c] <= e < in_ptr[c + 1])
    const int32_t* in_idx;     // [E] parent of edge e
    const int32_t* node_k;
    const double* npi_init;
    double* state;
    const uint8_t* frz;
    uint8_t frz_mark;
    const int32_t* eperm;      // CSR edge id -> message record
    const int32_t* nperm;      // node id -> slot of its vectors
};
// several evidence sets of a batch in one launch (blockIdx.y = the set): a launch per set cost a batch of 16 sets 0.1 ms per call
struct DagEvidenceBatch { DagEvidenceArgs set[kDagMaxSets]; };
struct DagInitBatch { DagInitArgs set[kDagMaxSets]; };
int launch_dag_init(const DagInitArgs& a, void* stream_handle);
int launch_dag_init_batch(const DagInitBatch& b, int n_sets, void* stream_handle);
int launch_bp_dag(const DagArgs& a, bool stream, void* stream_handle);
int launch_dag_evidence(const DagEvidenceArgs& a, void* stream_handle);
int launch_dag_evidence_batch(const DagEvidenceBatch& b, int n_sets, void* stream_handle);

// where things live in DagArgs::state, in double2 units (a record = 4 doubles = two of them)
__host__ __device__ inline int64_t dag_off_pim(int64_t E, int64_t, int par, int64_t e) { return (int64_t(par) * E + e) * 2; }
__host__ __device__ inline int64_t dag_off_lam(int64_t E, int64_t, int par, int64_t e) { return 4 * E + (int64_t(par) * E + e) * 2; }
__host__ __device__ inline int64_t dag_off_npi(int64_t E, int64_t n, int par, int64_t v) { return 8 * E + (int64_t(par) * n + v) * 2; }
__host__ __device__ inline int64_t dag_off_nlam(int64_t E, int64_t n, int par, int64_t v) { return 8 * E + 4 * n + (int64_t(par) * n + v) * 2; }
__host__ __device__ inline int64_t dag_state_doubles(int64_t E, int64_t n) { return 16 * E + 16 * n; }

}  // namespace bnmi

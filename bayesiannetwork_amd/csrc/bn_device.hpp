// bn_device.hpp -- structs shared by the host engine and the HIP kernels.
#pragma once

#include <cstdint>

#include "bn_plan.hpp"

namespace bnmi {

#ifndef BN_BLOCK_THREADS
#define BN_BLOCK_THREADS 256
#endif
constexpr int kBlockThreads = BN_BLOCK_THREADS;  // 8 tile waves per CU at 242 VGPRs whatever the block size
constexpr int kWavesPerBlock = kBlockThreads / kWave;

// Device-resident control block.  A run is identified by its run id (1, 2, ... per engine): the
// run is over for the kernels once done_run == the id their launch carries, so nothing has to be
// reset between runs (a stale block belongs to an older id).
struct Ctl {
    uint32_t done_run;  // id of the last run whose sweeps decided the stop themselves (maximum_difference < eps)
    int32_t done;       // host copy only: 0 running, 1 converged, 2 stopped at max_sweeps
    int32_t n_sweeps;   // iterations of the reference's while(true) loop that were executed
    uint32_t run_id;    // host copy only: the run the other fields describe
    double last_res;    // maximum_difference of the last executed sweep
    unsigned long long t_first;  // wall_clock64() (100 MHz) when sweep 0 of the run started
    unsigned long long t_last;   // ... when the launch after the last executed sweep started
};

struct BpBuffers {
    const TileDesc* tiles;
    const ClassDesc* classes;
    const FlatEntry* flat_tab;  // per-entry digits / summation places of the ordered any-arity classes
    int32_t n_tiles;
    const double* cpt;
    double* rec0;        // double-buffered message records (no arrays here: a dynamically indexed
    double* rec1;        //   kernarg array forces the whole struct into scratch memory)
    double* node0;       // double-buffered pi / lambda node vectors
    double* node1;
    const MsgRef* out_refs;
    const MsgRef* in_refs;
    uint8_t* frozen;     // per lane-slot evidence marker (preconditional_node_, :69): marked when == frozen_mark
    uint8_t frozen_mark; // the value that marks the evidence set in force (a new set takes the next value: nothing to clear)
    const int32_t* slot_node;
    const int64_t* slot_boff;
    const int32_t* node_tile;
    const int32_t* node_nl;
    // exchange region of the record buffers (bn_plan.hpp), double2 units
    int64_t g_base, seg_d2, seg_data_d2;
    int64_t rec_total_doubles;  // size of one record buffer, exchange region included
    int32_t rank, nranks;
    double* res_hist;
    int32_t res_cap;
    Ctl* ctl;
    double* beliefs;
};

// Several evidence sets in one launch of the per-sweep kernels (bn_bp_run_batch on networks the resident
// kernel does not cover): blockIdx.y = evidence set; set q's records, node vectors, marks, beliefs,
// residual history and control block follow set 0's at these strides (elements of the respective array).
struct SetStrides {
    int64_t rec, node, slot, belief;
    int32_t res_hist;
};
__host__ __device__ inline void shift_to_set(BpBuffers& b, const SetStrides& st, int set) {
    b.rec0 += set * st.rec; b.rec1 += set * st.rec;
    b.node0 += set * st.node; b.node1 += set * st.node;
    b.frozen += set * st.slot;
    b.beliefs += set * st.belief;
    b.res_hist += int64_t(set) * st.res_hist;
    b.ctl += set;
}

struct SweepArgs {
    BpBuffers b;
    const double* rec_in;   // buffer (sweep & 1): the state this iteration reads
    double* rec_out;        // buffer ((sweep+1) & 1): the reference's new_* maps
    const double* node_in;
    double* node_out;
    double eps;
    int32_t sweep;       // 0-based index of this iteration
    int32_t tile_begin;  // this launch covers tiles [tile_begin, tile_end) ...
    int32_t tile_end;
    int32_t book;        // ... and, if set, the residual bookkeeping (done by the first wave past tile_end)
    uint32_t run_id;
    SetStrides sets;     // batched launches only (gridDim.y = number of evidence sets)
};

struct FinishArgs {
    BpBuffers b;
    double eps;
    int32_t sweeps_launched;
    int32_t final_batch;  // 1: max_sweeps reached with this batch -> stop even if not converged
    uint32_t run_id;
    Ctl* host_ctl;        // pinned host copy of Ctl, written by the kernel itself (no D2H copy command)
    SetStrides sets;      // gridDim.y > 1: one evidence set per y (host_ctl is an array then)
};

struct EvidenceArgs {
    BpBuffers b;
    int32_t ne;
    const int32_t* ev_node;
    const int32_t* ev_off;
    const double* ev_val;
};

// The whole run in one launch with every tile resident in registers (bn_resident.hip).
constexpr int kResidentWaves = 8;       // 512 threads per block, one tile per wave, <= 256 VGPRs
constexpr int kResidentLdsSlots = 18;   // double2 slots per lane of CPT kept in LDS (36 of the 64 entries of a k = 4, two-parent table)
constexpr int kResidentMaxSets = 4;     // evidence sets one launch walks round-robin (bn_bp_run_batch); measured on the 316x316 grid:
                                        // 9.3 / 9.0 / 11.3 us per set-sweep with 2 / 4 / 8 sets per launch
constexpr int kResidentBudget = 1024;   // iterations one launch may execute (size of ResidentSync::res)
constexpr int kResidentMaxBlocks = 256;
struct ResidentSync {                   // one per evidence set; zeroed at creation, after an aborted launch and when the
                                        // generation counter would wrap (generations count on across launches)
    unsigned abort;                     // (set 0's) non-zero: a bounded wait gave up
    unsigned pad_[31];
    struct Group {
        unsigned gen;                   // generation | verdict << 30 of the group's last released barrier
        unsigned pad_[31];
    } grp[8];
    unsigned long long blk[kResidentMaxBlocks][2];  // per tile block: the same, written by the block
    unsigned long long res[kResidentBudget];        // per-iteration maximum_difference, bit patterns (service block)
    unsigned long long blk_odd[kResidentMaxBlocks][2];  // direct form: the granules of odd iterations (even ones in blk)
};
// Dataflow form (single evidence set, more than one tile block): no grid barrier.  Every TILE (wave) publishes, per
// iteration, a pair of 8-byte granules {generation | half of its residual's bit pattern} into the slot of the
// iteration's parity; a tile starts its next iteration once the tiles it exchanges messages with (Plan::nbr) carry
// the generation of the previous one.  The stop decision lags by one iteration: a service block collects the
// granules of iteration i while the tiles already compute i + 1, and publishes {generation, verdict}; a tile starts
// iteration i + 2 only once the verdict of i is known, so records and node vectors of the state the run stops in
// are still intact when it does (double buffers: the speculative iteration writes the OTHER buffer).
constexpr int kFlowMaxTiles = kResidentWaves * kResidentMaxBlocks;
static_assert(kFlowMaxTiles == kFlowSlotsPerRank, "one granule slot per possible tile of a rank");
constexpr int kMaxRanks = 16;          // ranks of a sharded engine that can exchange inside the resident kernel
enum : unsigned { kFlowGoOn = 0, kFlowConverged = 1, kFlowCapped = 2, kFlowAbort = 3, kFlowBudget = 4 };
// One allocation per engine: this header, then the granule table  [2 iteration parities][nranks * kFlowSlotsPerRank][2]
// of {generation << 32 | residual half}: slot rank * kFlowSlotsPerRank + tile.  A rank writes its own tiles' slots --
// in its own table and, for tiles that touch a cut edge, in the tables of the ranks across the cut (peer-mapped
// memory: hipIpc handles between processes, plain pointers inside one) -- and polls only its own table.
// Zeroed at creation, after an aborted launch and before the generation would wrap.
struct FlowSync {
    struct Line {
        unsigned long long word;        // low 32 bits: generation of the last decided iteration; high 32: its verdict (kFlow*)
        unsigned long long pad_[15];
    } verdict[8];                       // copies on lines of their own: tile blocks poll copy blockIdx % 8
    unsigned long long rank_granule[2][kMaxRanks][2];  // [parity][rank]: every rank's {generation | residual half} pair, written by that rank's service block
    unsigned long long res[kResidentBudget];           // per-iteration maximum_difference, bit patterns
};
__host__ __device__ inline size_t flow_sync_bytes(int nranks) {
    return sizeof(FlowSync) + sizeof(unsigned long long) * 2 * size_t(nranks) * kFlowSlotsPerRank * 2;
}
// What a rank needs of another to exchange inside the kernel (device array [nranks]; entry `rank` is the engine itself)
struct PeerTable {
    FlowSync* flow;   // the peer's sync block
    double* rec0;     // the peer's record buffers: cut-edge halves are stored here as well as locally
    double* rec1;
    int64_t g_base;   // where the exchange region starts in THEIR record buffers (double2 units): the region's layout is
                      // the same on every rank, its place behind the rank's own tile records is not
    int64_t rec_bytes;  // size of one of their record buffers
};
struct ResidentArgs {
    BpBuffers b;          // evidence set 0; set q's buffers follow at the strides below
    double eps;
    int32_t max_sweeps;   // 0 = unbounded like the reference
    int32_t sweep_begin;  // first iteration of this launch
    int32_t budget;       // iterations this launch may execute (<= kResidentBudget)
    uint32_t run_id;
    uint32_t gen_base;    // generations used by earlier launches on these sync blocks (< 2^29)
    unsigned long long timeout_ticks;  // bound of one barrier wait, 100 MHz ticks
    ResidentSync* sync;   // [n_sets]
    Ctl* host_ctl;        // [n_sets], pinned
    int32_t n_tile_blocks;  // blocks [0, n_tile_blocks) carry tiles; the launch's further blocks serve the barrier
    int32_t waves;          // tiles (= waves) per block: kResidentWaves, or half of it on networks small enough (one wave per SIMD)
    int32_t n_sets;       // 1: node vectors stay in registers; > 1: they go through memory between a set's turns
    uint32_t set_mask;    // sets still running (a continued launch skips the others)
    int64_t rec_stride, node_stride;  // doubles between consecutive sets' record / node buffers
    int64_t slot_stride;              // bytes between their evidence marks
    int64_t belief_stride;            // doubles between their beliefs
    int32_t res_hist_stride;          // doubles between their residual histories
    // dataflow form (flow != nullptr; n_sets == 1, n_tile_blocks > 1)
    FlowSync* flow;
    // ... of a sharded engine (peers != nullptr): halo exchange inside the kernel
    const PeerTable* peers;           // [b.nranks]
    const uint32_t* pub_mask;         // [n_tiles] bit q: the tile has a neighbour tile on rank q (gets its granules too)
    int32_t n_interior;               // tiles [0, n_interior) touch no cut edge
    const int32_t* nbr;               // [n_tiles][nbr_chunks][kWave] neighbour tiles, -1 padded (Plan::nbr)
    int32_t nbr_chunks;
    int32_t poll_sleep;               // pause between two polls of a waiting tile, in units of s_sleep(8) = 512 cycles
    unsigned* host_abort;             // pinned: set by whoever gives up a bounded wait
    int32_t direct;                   // grid-barrier form, one evidence set: every tile block reads all blocks' granules itself (no service block)
    int32_t first_poll_delay;         // direct form: margin, in 10 ns ticks, between the predicted arrival of the last block and a block's first poll
};
int launch_bp_resident(const ResidentArgs& a, int grid_blocks, int lean_k, void* stream);  // a.flow != nullptr: the dataflow form  // lean_k: uniform arity with <= 2 children per node, else 0

// launchers (bn_kernels.hip)
int launch_bp_evidence(const EvidenceArgs& a, void* stream);  // bn_bp_set_evidence: marks + vectors
int launch_bp_reset(const BpBuffers& b, void* stream);         // residual slots; after an abnormal end only
// n_sets > 1: the batched instantiation, one evidence set per blockIdx.y (a.sets valid)
int launch_bp_sweep(const SweepArgs& a, int grid_blocks, int n_sets, bool nontemporal, bool light, int variants, void* stream);
// per-variant-set instantiations (bn_sweep_*.hip)
int launch_bp_sweep_u(const SweepArgs& a, int grid_blocks, int n_sets, bool nontemporal, void* stream);
int launch_bp_sweep_ug(const SweepArgs& a, int grid_blocks, int n_sets, bool nontemporal, void* stream);
int launch_bp_sweep_all(const SweepArgs& a, int grid_blocks, int n_sets, void* stream);
int launch_bp_finish(const FinishArgs& a, int grid_blocks, int n_sets, void* stream);

}  // namespace bnmi

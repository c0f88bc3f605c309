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

// Device-resident control block of one BP run.
struct Ctl {
    int32_t done;       // 0 running, 1 converged (maximum_difference < eps), 2 stopped at max_sweeps
    int32_t n_sweeps;   // iterations of the reference's while(true) loop that were executed
    double last_res;    // maximum_difference of the last executed sweep
    uint32_t p_abort;   // persistent kernel: non-zero = a bounded wait gave up (copied by the finish kernel)
    uint32_t p_conv;    // persistent kernel: iteration count at which it ended, 0 = it did not end
};

struct PersistSync;

struct BpBuffers {
    const TileDesc* tiles;
    const ClassDesc* classes;
    int32_t n_tiles;
    const double* cpt;
    double* rec0;        // double-buffered message records (no arrays here: a dynamically indexed
    double* rec1;        //   kernarg array forces the whole struct into scratch memory)
    double* node0;       // double-buffered pi / lambda node vectors
    double* node1;
    const MsgRef* out_refs;
    const MsgRef* in_refs;
    uint8_t* frozen;     // per lane-slot evidence marker (preconditional_node_, :69)
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

struct SweepArgs {
    BpBuffers b;
    const double* rec_in;   // buffer (sweep & 1): the state this iteration reads
    double* rec_out;        // buffer ((sweep+1) & 1): the reference's new_* maps
    const double* node_in;
    double* node_out;
    double eps;
    int32_t sweep;       // 0-based index of this iteration
    int32_t book_tile;   // first wave index past the tiles: it does the residual bookkeeping
};

struct FinishArgs {
    BpBuffers b;
    double eps;
    int32_t sweeps_launched;
    int32_t final_batch;  // 1: max_sweeps reached with this batch -> stop even if not converged
    int32_t ne;           // evidence marks to clear once the run is over
    const int32_t* ev_node;
    const struct PersistSync* psync;  // persistent run: status to fold into Ctl (else nullptr)
    Ctl* host_ctl;        // pinned host copy of Ctl, written by the kernel itself (no D2H copy command)
};

struct EvidenceArgs {
    BpBuffers b;
    int32_t ne;
    const int32_t* ev_node;
    const int32_t* ev_off;
    const double* ev_val;
};

// Synchronisation block of the persistent dataflow kernel (bn_persist.hip); zeroed before each run.
struct PersistSync {
    unsigned completed;  // iterations every tile has finished
    unsigned conv;       // iteration count at which the run ended (converged / capped), 0 = running
    unsigned abort;      // non-zero: a bounded wait gave up (1 global slack, 2 neighbour flag, 3 bad tile)
    unsigned pad_;
    unsigned count[4];   // arrivals per iteration (mod 4)
};

struct PersistArgs {
    BpBuffers b;
    const int32_t* nbr_ptr;         // tile adjacency (bn_plan.hpp)
    const int32_t* nbr_idx;
    PersistSync* sync;
    unsigned* flags;                // [n_tiles] iterations finished by each tile
    unsigned long long* res_tile;   // [4][n_tiles] per-tile residual bit patterns, by iteration mod 4
    double eps;
    int32_t max_sweeps;
    int32_t n_tiles;
    unsigned long long timeout_ticks;  // 100 MHz ticks
};

int launch_bp_persistent(const PersistArgs& a, int grid_blocks, void* stream);

// launchers (bn_kernels.hip)
int launch_bp_begin(const EvidenceArgs& a, void* stream);
int launch_bp_sweep(const SweepArgs& a, int grid_blocks, bool nontemporal, bool light, void* stream);
int launch_bp_finish(const FinishArgs& a, int grid_blocks, void* stream);

}  // namespace bnmi

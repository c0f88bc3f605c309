// bn_device.hpp -- structs shared by the host engine and the HIP kernels.
#pragma once

#include <cstdint>

#include "bn_plan.hpp"

namespace bnmi {

constexpr int kBlockThreads = 256;
constexpr int kWavesPerBlock = kBlockThreads / kWave;

// Device-resident control block of one BP run.
struct Ctl {
    int32_t done;       // 0 running, 1 converged (maximum_difference < eps), 2 stopped at max_sweeps
    int32_t n_sweeps;   // iterations of the reference's while(true) loop that were executed
    double last_res;    // maximum_difference of the last executed sweep
};

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
};

struct EvidenceArgs {
    BpBuffers b;
    int32_t ne;
    const int32_t* ev_node;
    const int32_t* ev_off;
    const double* ev_val;
};

// launchers (bn_kernels.hip)
int launch_bp_begin(const EvidenceArgs& a, void* stream);
int launch_bp_sweep(const SweepArgs& a, int grid_blocks, bool nontemporal, void* stream);
int launch_bp_finish(const FinishArgs& a, int grid_blocks, void* stream);

}  // namespace bnmi

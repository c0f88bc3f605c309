// bn_kernels.hip -- hand-written gfx950 kernels for synchronous (Jacobi) loopy belief propagation,
// one launch per sweep.
//
// One launch = one iteration of the reference's while(true) loop
// (bayesian/inference/belief_propagation.hpp:75-148): message phase (:78-88), node phase
// (:91-101), residual (:105-131) and commit (:135-143, here a buffer swap) are fused into a
// single pass in which every node's CPT is read exactly once.  Work decomposition: one 64-lane
// wavefront per tile (bn_plan.hpp); the tile code is in bn_tiles.hpp.
//
// Iteration 0 reads nothing but the CPT and the evidence: the reference's initial state (:33-73)
// is all-ones messages, pi = lambda = 1 (roots: their CPT row), so it is synthesised in registers
// instead of being written to HBM by an initialisation pass and read back.
//
// Convergence is decided on the device: sweep s accumulates max|new-old| over messages into a
// ring of 256 slots (atomic umax on the bit pattern of a non-negative double); ONE extra wave of
// launch s+1 reduces them and marks the run done once maximum_difference < eps (:147); every later
// launch returns at once, so the host enqueues launches ahead without synchronising per sweep.
#include "bn_sweep.hpp"

namespace bnmi {

// The same iteration for networks WITHOUT register-resident tiles (any-arity and one-lane tiles
// only): those tiles need a third of the registers and are latency-bound, so this instantiation
// runs at twice the occupancy (4 waves per SIMD).
template <bool BATCH>
__global__ __launch_bounds__(kBlockThreads, 4) void bp_sweep_light_kernel(SweepArgs a_in) {
    const SweepArgs a = BATCH ? sweep_args_of_set(a_in) : a_in;
    __shared__ double flat_lds[kWavesPerBlock][kFlatLds];
    const BpBuffers& b = a.b;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int done = __hip_atomic_load(&b.ctl->done_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.run_id;
    const int tile = a.tile_begin + logical_block() * kWavesPerBlock + wave;
    if (tile >= a.tile_end) {
        if (done == 0 && a.book != 0 && tile == a.tile_end) sweep_bookkeeping(a, lane);
        return;
    }
    const IO io{a.rec_in, a.rec_out, a.node_in, a.node_out, a.sweep == 0};
#ifdef BN_TILE_CLOCK
    const unsigned long long t_entry = wall_clock64();
#endif
    const TileDesc td = b.tiles[tile];
    if (done != 0) return;
    const double wres = run_tile_light(b, io, td, lane, flat_lds[wave]);
    publish_residual(b, a.rec_out, tile, wres, lane);
#ifdef BN_TILE_CLOCK
    if (lane == 0 && td.slot_base < kTileClockTiles) {
        g_tile_clock[td.slot_base][9] = t_entry;
        g_tile_clock[td.slot_base][11] = wall_clock64();
    }
#endif
}

// bn_bp_set_evidence: apply the evidence (belief_propagation.hpp:68-73) to the nodes this rank owns:
// pi(v) = lambda(v) = the given vector in the buffer iteration 0 reads, node marked
// (preconditional_node_: the slot takes the mark value of this evidence set, so the previous set's marks need no
// clearing).  The arrays are read straight from the caller-side staging block in page-locked host memory.  The marks and vectors stay
// until the next bn_bp_set_evidence: a marked node's vectors are copied forward by every sweep, so
// both buffers keep holding them and any number of runs can follow without touching them again.
__global__ __launch_bounds__(kBlockThreads) void bp_evidence_kernel(EvidenceArgs a) {
    const BpBuffers& b = a.b;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.ne) return;
    const int v = a.ev_node[j];
    const int t = b.node_tile[v];
    if (t < 0) return;  // owned by another rank
    const TileDesc td = b.tiles[t];
    const int nl = b.node_nl[v];
    const int half = ((td.kv + 1) & ~1) >> 1;
    for (int i = 0; i < td.kv; ++i) {
        const double x = a.ev_val[a.ev_off[j] + i];
        b.node0[td.node_base + vidx(0, i, td.npt, nl)] = x;
        b.node0[td.node_base + vidx(half, i, td.npt, nl)] = x;
        b.node1[td.node_base + vidx(0, i, td.npt, nl)] = x;
        b.node1[td.node_base + vidx(half, i, td.npt, nl)] = x;
    }
    b.frozen[td.slot_base + nl] = b.frozen_mark;
}

// This rank's residual slots in both buffers.  A finished run leaves them zero itself
// (bp_finish_kernel); this kernel runs only after a run that did not end normally.
__global__ __launch_bounds__(kBlockThreads) void bp_reset_kernel(BpBuffers b) {
    unsigned long long* r0 = res_row(b, b.rec0, b.rank);
    unsigned long long* r1 = res_row(b, b.rec1, b.rank);
    for (int q = threadIdx.x; q < kResSlots; q += kBlockThreads) { r0[q] = 0ull; r1[q] = 0ull; }
}

// After a batch of sweeps (and their exchanges).  Wave 0 of block 0 settles the run: if no sweep
// launch marked it done, it reduces the last launched sweep's residual and decides (converged /
// max_sweeps reached / go on), reports to the pinned host block, and -- once the run is over -- leaves
// the residual slots zero for the next run.  Every other wave writes belief = normalize(pi % lambda)
// (:151-158) for its tile's nodes from the buffer the last executed sweep wrote; when the host has to
// go on (predicted sweep count too low) those beliefs are simply overwritten by the next finish.
__global__ __launch_bounds__(kBlockThreads) void bp_finish_kernel(FinishArgs a) {
    BpBuffers b = a.b;
    Ctl* host_ctl = a.host_ctl;
    if (gridDim.y > 1) {  // batched run: one evidence set per y
        shift_to_set(b, a.sets, blockIdx.y);
        host_ctl += blockIdx.y;
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    // Marked by a sweep launch = a previous kernel, or by wave 0 of this one (below, which releases the fields it wrote
    // before the mark).  The mark and then, behind it, the sweep count are read with L1-bypassing loads: what the
    // release made visible in L2 is what they see.  (An ACQUIRE load here invalidated the CU's L1 once per wave --
    // 65 536 times on the 2048x2048 grid -- and held this kernel at ~0.8 TB/s: 595 us there, 17 us on the 316x316 grid.)
    const bool marked = __hip_atomic_load(&b.ctl->done_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.run_id;
    int n_sweeps = a.sweeps_launched;
    if (marked) n_sweeps = __hip_atomic_load(&b.ctl->n_sweeps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0 && wave == 0) {
        int done = 1;
        double r = marked ? __hip_atomic_load(&b.ctl->last_res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        unsigned long long t_last = marked ? __hip_atomic_load(&b.ctl->t_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : wall_clock64();
        if (marked && !(r < a.eps)) done = 2;  // marked by an earlier finish of this run that stopped it at max_sweeps
        if (!marked) {  // sweep (launched-1) wrote buffer (launched & 1)
            r = reduce_residual(b, (a.sweeps_launched & 1) ? b.rec1 : b.rec0, lane);
            if (lane == 0 && a.sweeps_launched - 1 < b.res_cap) b.res_hist[a.sweeps_launched - 1] = r;
            done = (r < a.eps) ? 1 : (a.final_batch ? 2 : 0);
        }
        if (lane == 0) {
            host_ctl->last_res = r; host_ctl->n_sweeps = n_sweeps;
            host_ctl->t_first = b.ctl->t_first; host_ctl->t_last = t_last;
            host_ctl->run_id = a.run_id; host_ctl->done = done;
            if (!marked && done != 0) {
                // the run ends here: mark it on the device too, so that launches a batched run still issues for
                // its other evidence sets leave this one alone
                b.ctl->n_sweeps = n_sweeps; b.ctl->last_res = r; b.ctl->t_last = t_last;
                __hip_atomic_store(&b.ctl->done_run, a.run_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (done != 0) {
            unsigned long long* r0 = res_row(b, b.rec0, b.rank);
            unsigned long long* r1 = res_row(b, b.rec1, b.rank);
#pragma unroll
            for (int q = 0; q < kResSlots / kWave; ++q) { r0[q * kWave + lane] = 0ull; r1[q * kWave + lane] = 0ull; }
        }
    }
    const int tile = blockIdx.x * kWavesPerBlock + wave;
    if (tile >= b.n_tiles) return;
    tile_beliefs(b, b.tiles[tile], (n_sweeps & 1) ? b.node1 : b.node0, lane);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : int(e); }

int launch_bp_evidence(const EvidenceArgs& a, void* stream) {
    if (a.ne <= 0) return 0;
    const int blocks = (a.ne + kBlockThreads - 1) / kBlockThreads;
    (void)hipGetLastError();  // drop any stale error of this thread
    hipLaunchKernelGGL(bp_evidence_kernel, dim3(blocks), dim3(kBlockThreads), 0, (hipStream_t)stream, a);
    return hip_rc(hipGetLastError());
}
int launch_bp_reset(const BpBuffers& b, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    hipLaunchKernelGGL(bp_reset_kernel, dim3(1), dim3(kBlockThreads), 0, (hipStream_t)stream, b);
    return hip_rc(hipGetLastError());
}
// variants: bit v set = the plan has tiles of variant v (bn_plan.hpp); light: any-arity / one-lane tiles only
int launch_bp_sweep(const SweepArgs& a, int grid_blocks, int n_sets, bool nontemporal, bool light, int variants, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    if (light) {
        if (n_sets > 1)
            hipLaunchKernelGGL(bp_sweep_light_kernel<true>, dim3(grid_blocks, n_sets), dim3(kBlockThreads), 0, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(bp_sweep_light_kernel<false>, dim3(grid_blocks), dim3(kBlockThreads), 0, (hipStream_t)stream, a);
        return hip_rc(hipGetLastError());
    }
    if (variants & ((1 << kVariantFlat) | (1 << kVariantGeneric))) return launch_bp_sweep_all(a, grid_blocks, n_sets, stream);
    if (variants & (1 << kVariantGroup)) return launch_bp_sweep_ug(a, grid_blocks, n_sets, nontemporal, stream);
    return launch_bp_sweep_u(a, grid_blocks, n_sets, nontemporal, stream);
}
int launch_bp_finish(const FinishArgs& a, int grid_blocks, int n_sets, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    hipLaunchKernelGGL(bp_finish_kernel, dim3(grid_blocks, n_sets > 1 ? n_sets : 1), dim3(kBlockThreads), 0, (hipStream_t)stream, a);
    return hip_rc(hipGetLastError());
}

}  // namespace bnmi

BN_TILE_CLOCK_GETTER(bn_debug_tile_clock_light)

// bn_sweep.hpp -- the per-sweep kernel template, instantiated per set of tile variants in
// bn_sweep_u.hip / bn_sweep_ug.hip / bn_sweep_all.hip (separate translation units: they compile in
// parallel, and a network made of register-resident tiles only runs a kernel that carries no code,
// registers or LDS for the other variants).
#pragma once

#include "bn_tiles.hpp"

namespace bnmi {

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// Logical block index with an XCD-contiguous mapping: hardware block b runs on XCD b % 8
// (observed, speed only), so logical chunk [x*nb/8, (x+1)*nb/8) of the tile list -- spatially
// adjacent tiles that share message records -- stays inside one XCD's L2.  gridDim.x % 8 == 0.
__device__ __forceinline__ int logical_block() {
    const int nb = gridDim.x, b = blockIdx.x;
    return (b & 7) * (nb >> 3) + (b >> 3);
}

// Residual bookkeeping of launch s, done by ONE wave that carries no tile: settle sweep s-1
// (record maximum_difference, mark the run done when it is < eps, :147), then zero this rank's slots
// in the buffer it just read -- sweep s+1 accumulates into them.  Tile waves never wait for it:
// a launch that starts after convergence only writes the buffer that is no longer current, and
// the launch after that sees the mark and returns at once.  Launch 0 has nothing to settle; the
// slots it accumulates into were left zero by the previous run's finish kernel.
__device__ __forceinline__ void sweep_bookkeeping(const SweepArgs& a, int lane) {
    const BpBuffers& b = a.b;
    unsigned long long* row = res_row(b, a.rec_in, b.rank);
    if (a.sweep == 0) {
        if (lane == 0) b.ctl->t_first = wall_clock64();
    } else {
        const double r = reduce_residual(b, a.rec_in, lane);
        if (lane == 0) {
            if (a.sweep - 1 < b.res_cap) b.res_hist[a.sweep - 1] = r;
            if (r < a.eps) {
                b.ctl->n_sweeps = a.sweep;
                b.ctl->last_res = r;
                b.ctl->t_last = wall_clock64();
                __hip_atomic_store(&b.ctl->done_run, a.run_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < kResSlots / kWave; ++q) row[q * kWave + lane] = 0ull;
}

// The launch's arguments moved to evidence set blockIdx.y (batched launches; everything stays wave-uniform).
__device__ __forceinline__ SweepArgs sweep_args_of_set(const SweepArgs& in) {
    SweepArgs a = in;
    const int set = blockIdx.y;
    shift_to_set(a.b, in.sets, set);
    a.rec_in += set * in.sets.rec; a.rec_out += set * in.sets.rec;
    a.node_in += set * in.sets.node; a.node_out += set * in.sets.node;
    return a;
}

// VARIANTS: bit kVariantUniform / kVariantGroup / kVariantFlat set = the plan has such tiles
// BATCH: gridDim.y evidence sets per launch, each with its own buffers, residual slots and done mark
template <bool NT, int VARIANTS, bool BATCH = false>
__global__ __launch_bounds__(kBlockThreads, 2) void bp_sweep_kernel(SweepArgs a_in) {
    const SweepArgs a = BATCH ? sweep_args_of_set(a_in) : a_in;
    constexpr bool FLAT = (VARIANTS >> kVariantFlat) & 1;
    // any-arity tiles only: staged terms, children's messages
    __shared__ double flat_lds[FLAT ? kWavesPerBlock : 1][FLAT ? kFlatLds : 1];
    const BpBuffers& b = a.b;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the run's done mark and the tile descriptor are fetched together: one round trip, not two, heads the chain
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
    const double wres = run_tile<NT, VARIANTS>(b, io, td, lane, flat_lds[FLAT ? wave : 0]);
    publish_residual(b, a.rec_out, tile, wres, lane);
#ifdef BN_TILE_CLOCK
    if (lane == 0 && td.slot_base < kTileClockTiles) {
        g_tile_clock[td.slot_base][9] = t_entry;
        g_tile_clock[td.slot_base][11] = wall_clock64();
    }
#endif
}


}  // namespace bnmi

// diagnostic builds only: every translation unit with sweep kernels has its own copy of the stamps and a getter for them
#ifdef BN_TILE_CLOCK
#define BN_TILE_CLOCK_GETTER(name)                                                                                          \
    extern "C" int name(unsigned long long* out, int n_tiles) {                                                             \
        if (n_tiles > bnmi::kTileClockTiles) n_tiles = bnmi::kTileClockTiles;                                               \
        return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(bnmi::g_tile_clock),                                                 \
                                       sizeof(unsigned long long) * bnmi::kTileClockStamps * n_tiles));                     \
    }
#else
#define BN_TILE_CLOCK_GETTER(name)
#endif

// bn_multi.hip -- all sweeps of a SMALL network in ONE launch of ONE workgroup.
//
// The per-sweep launch (bn_kernels.hip) costs a kernel boundary, a trip through the residual slots in
// memory and a launch floor of several microseconds per iteration of the reference's while(true) loop
// (belief_propagation.hpp:75-148).  A network of a few dozen tiles (the reference's own test graphs,
// ALARM-sized networks: BASELINE.json configs[0]) is nothing but that overhead.  Here one workgroup
// runs the whole loop: wave w owns tiles w, w + WAVES, ...; an iteration is the same tile code
// (bn_tiles.hpp, so the results are bit-identical to the launch path) followed by ONE __syncthreads(),
// which is at once
//   * the commit (:135-143): all of the block's waves sit on one CU and share its vector L1, stores
//     are write-through and the barrier waits for them, so the next iteration reads what this one wrote;
//   * the residual reduction (:105-131): each wave leaves max|new - old| of its tiles in an LDS slot,
//     every thread then reduces the slots itself and takes the SAME stop decision (:147) -- no atomics,
//     no memory round trip, no host.
// The launch also applies nothing and finishes everything: evidence is in place since
// bn_bp_set_evidence, the beliefs (:151-158) are written before the kernel ends, the outcome goes
// straight into the pinned host block.
//
// A launch executes at most `budget` iterations (a run that needs more is continued by another
// launch from where this one stopped), so an evidence set on which loopy BP does not converge -- the
// reference would spin forever -- cannot keep the GPU inside one kernel.
#include "bn_tiles.hpp"

namespace bnmi {

template <int WAVES, bool LIGHT>
__global__ __launch_bounds__(WAVES * kWave) void bp_multi_kernel(MultiArgs a) {
    __shared__ double flat_lds[WAVES][kFlatLds];
    __shared__ unsigned long long res_slot[2][16];
    const BpBuffers& b = a.b;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t_first = wall_clock64();
    int n_done = a.sweep_begin, done = 0;
    double r_last = 0.0;
    for (int it = 0; it < a.budget; ++it) {
        const int s = a.sweep_begin + it;
        const bool odd = (s & 1) != 0;
        const IO io{odd ? b.rec1 : b.rec0, odd ? b.rec0 : b.rec1, odd ? b.node1 : b.node0, odd ? b.node0 : b.node1, s == 0};
        double wres = 0.0;
        for (int tile = wave; tile < b.n_tiles; tile += WAVES) {
            const TileDesc td = b.tiles[tile];
            double r;
            if constexpr (LIGHT) r = run_tile_light(b, io, td, lane, flat_lds[wave]);
            else r = run_tile<false, kVarAll>(b, io, td, lane, flat_lds[wave]);
            wres = res_acc(wres, r);
        }
        const unsigned long long bits = wave_umax((unsigned long long)__double_as_longlong(wres));
        if (lane == 0) res_slot[s & 1][wave] = bits;
        __syncthreads();
        unsigned long long m = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const unsigned long long x = res_slot[s & 1][w];
            m = x > m ? x : m;
        }
        double r = __longlong_as_double((long long)m);
        r = r < DBL_MIN ? DBL_MIN : r;  // maximum_difference starts at numeric_limits<double>::min() (:105)
        if (threadIdx.x == 0 && s < b.res_cap) b.res_hist[s] = r;
        n_done = s + 1;
        r_last = r;
        if (r < a.eps) { done = 1; break; }                                   // strict '<' (:147)
        if (a.max_sweeps > 0 && n_done >= a.max_sweeps) { done = 2; break; }
    }
    const unsigned long long t_last = wall_clock64();
    if (done != 0) {
        const double* node_buf = (n_done & 1) ? b.node1 : b.node0;
        for (int tile = wave; tile < b.n_tiles; tile += WAVES) tile_beliefs(b, b.tiles[tile], node_buf, lane);
    }
    if (threadIdx.x == 0) {
        a.host_ctl->last_res = r_last; a.host_ctl->n_sweeps = n_done;
        a.host_ctl->t_first = t_first; a.host_ctl->t_last = t_last;
        a.host_ctl->run_id = a.run_id; a.host_ctl->done = done;
    }
}

int launch_bp_multi(const MultiArgs& a, bool light, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    if (light)
        hipLaunchKernelGGL((bp_multi_kernel<kMultiWavesLight, true>), dim3(1), dim3(kMultiWavesLight * kWave), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((bp_multi_kernel<kMultiWaves, false>), dim3(1), dim3(kMultiWaves * kWave), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

}  // namespace bnmi

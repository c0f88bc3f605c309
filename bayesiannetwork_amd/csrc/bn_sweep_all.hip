// bn_sweep_all.hip -- instantiation of the per-sweep kernel (bn_sweep.hpp) for every tile variant.
#include "bn_sweep.hpp"

namespace bnmi {

// (working sets beyond the Infinity Cache made of any-arity tiles are latency-bound either way: plain stores)
int launch_bp_sweep_all(const SweepArgs& a, int grid_blocks, int n_sets, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    if (n_sets > 1)  // one evidence set per blockIdx.y; plain stores (a batch is sized for the Infinity Cache or latency-bound)
        hipLaunchKernelGGL((bp_sweep_kernel<false, kVarAll, true>), dim3(grid_blocks, n_sets), dim3(kBlockThreads), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((bp_sweep_kernel<false, kVarAll>), dim3(grid_blocks), dim3(kBlockThreads), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

}  // namespace bnmi

BN_TILE_CLOCK_GETTER(bn_debug_tile_clock)

// bn_sweep_u.hip -- instantiation of the per-sweep kernel (bn_sweep.hpp) for register-resident tiles only.
#include "bn_sweep.hpp"

namespace bnmi {

int launch_bp_sweep_u(const SweepArgs& a, int grid_blocks, int n_sets, bool nontemporal, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    if (n_sets > 1)  // one evidence set per blockIdx.y; plain stores (a batch is sized for the Infinity Cache or latency-bound)
        hipLaunchKernelGGL((bp_sweep_kernel<false, kVarU, true>), dim3(grid_blocks, n_sets), dim3(kBlockThreads), 0, (hipStream_t)stream, a);
    else if (nontemporal)
        hipLaunchKernelGGL((bp_sweep_kernel<true, kVarU>), dim3(grid_blocks), dim3(kBlockThreads), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((bp_sweep_kernel<false, kVarU>), dim3(grid_blocks), dim3(kBlockThreads), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

}  // namespace bnmi

BN_TILE_CLOCK_GETTER(bn_debug_tile_clock_u)

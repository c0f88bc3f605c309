// bn_fit_kernels.hip -- maximum-likelihood CPT fitting from a table of joint patterns,
// reference bayesian/sampler.hpp:81-163 (sampler::make_cpt): for every node and every parent
// assignment, count[state] = sum of the occurrence counts of the patterns that show that
// assignment and that state; the CPT row is count / sum(count), or uniform when no pattern
// shows the assignment (:140-151).  Counting is integer work (exact); one block per node, the
// node's counters in LDS when the CPT has <= 4096 entries, patterns streamed from a
// [node][pattern] byte matrix so that every wave load is contiguous.
#include <hip/hip_runtime.h>

#include "bn_fit.hpp"

namespace bnmi {

__global__ __launch_bounds__(256) void fit_count_kernel(FitArgs a) {
    __shared__ unsigned long long sh[kFitLdsEntries];
    const int v = blockIdx.x;
    const int kv = a.k[v];
    const int e0 = a.in_ptr[v], m = a.in_ptr[v + 1] - e0;
    const int64_t coff = a.cpt_off[v];
    const int64_t csz = a.cpt_off[v + 1] - coff;
    const bool in_lds = csz <= kFitLdsEntries;
    unsigned long long* cnt = a.counts + coff;
    if (in_lds)
        for (int q = threadIdx.x; q < csz; q += blockDim.x) sh[q] = 0ull;
    __syncthreads();
    for (int64_t p = threadIdx.x; p < a.n_patterns; p += blockDim.x) {
        int64_t row = 0;  // parent assignment, first parent most significant
        for (int j = 0; j < m; ++j) {
            const int u = a.in_idx[e0 + j];
            row = row * a.k[u] + a.patterns[int64_t(u) * a.n_patterns + p];
        }
        const int st = a.patterns[int64_t(v) * a.n_patterns + p];
        const unsigned long long w = a.weights[p];
        if (in_lds) atomicAdd(&sh[row * kv + st], w);
        else atomicAdd(&cnt[row * kv + st], w);
    }
    __syncthreads();
    if (in_lds)
        for (int q = threadIdx.x; q < csz; q += blockDim.x) cnt[q] = sh[q];
}

// one thread per CPT row: count / row total, uniform when the row was never observed
__global__ void fit_normalize_kernel(FitArgs a) {
    const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r >= a.n_rows) return;
    const int v = a.row_node[r];
    const int kv = a.k[v];
    const int64_t base = a.row_off[r];
    unsigned long long total = 0;
    for (int i = 0; i < kv; ++i) total += a.counts[base + i];
    const double parameter = double(total);
    for (int i = 0; i < kv; ++i)
        a.cpt_out[base + i] = (total == 0) ? 1.0 / kv : double(a.counts[base + i]) / parameter;
}

int launch_fit(const FitArgs& a, void* stream) {
    (void)hipGetLastError();
    if (a.n > 0) hipLaunchKernelGGL(fit_count_kernel, dim3(a.n), dim3(256), 0, (hipStream_t)stream, a);
    if (a.n_rows > 0)
        hipLaunchKernelGGL(fit_normalize_kernel, dim3(unsigned((a.n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

}  // namespace bnmi

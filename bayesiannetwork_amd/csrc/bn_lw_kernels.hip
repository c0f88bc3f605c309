// bn_lw_kernels.hip -- likelihood weighting, reference bayesian/inference/likelihood_weighting.hpp.
//
// One thread draws kLwPerThread ancestral samples; a block walks the nodes in topological order
// together, so the node, its CPT (staged once per block in LDS) and its evidence flag are
// block-uniform.  Per node and sample (weighted_sample, :122-173):
//   row   = parent assignment, first parent most significant, from the state matrix
//   evidence node  : w *= cpt[row][ev], state = ev                         (:148-153)
//   otherwise      : state = first i with cum_{i-1} <= u < cum_i, else k-1   (:154-158, :177-193)
// (rejection / logic sampling, reference rejection_sampling.hpp:33-167, is the same walk with the
// evidence nodes sampled like any other and w = 1 if every one of them came out as observed, else 0)
// and hist[v][state] += w (:45-49), in a second kernel once w is final, pre-reduced per thread, per wave and per block before one
// fp64 atomicAdd per (block, node, state).  Uniforms come from Philox4x32-10 keyed by the seed
// and indexed by (global sample id, topological position) -- see oracle/lw_oracle.c for the
// exact mapping, which this kernel reproduces bit for bit, so sampled states are identical.
#include <hip/hip_runtime.h>

#include "bn_lw.hpp"

namespace bnmi {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__global__ __launch_bounds__(kLwThreads) void lw_sample_kernel(LwArgs a) {
    __shared__ double sh_cpt[kLwLdsDoubles];
    const int tid = threadIdx.x;
    const uint64_t local0 = uint64_t(blockIdx.x) * kLwBlockSamples + tid;  // + r * kLwThreads
    const bool reject = a.mode == 1;
    double w[kLwPerThread];
    uint32_t rnd[kLwPerThread][4];
#pragma unroll
    for (int r = 0; r < kLwPerThread; ++r) w[r] = 1.0;  // :124
    const uint32_t key0 = uint32_t(a.seed), key1 = uint32_t(a.seed >> 32);

    for (int t = 0; t < a.n; ++t) {
        const int v = a.topo[t];
        const int kv = a.k[v];
        const int e0 = a.in_ptr[v], e1 = a.in_ptr[v + 1];
        const int64_t coff = a.cpt_off[v];
        const int64_t csz = a.cpt_off[v + 1] - coff;
        const int ev = a.ev_state[v];
        const bool draws = reject || ev < 0;  // logic sampling draws evidence nodes too
#ifdef BN_LW_NOLDS
        const bool in_lds = false;
#else
        const bool in_lds = csz <= kLwLdsDoubles;
#endif
        __syncthreads();  // previous node's LDS reads are finished
        if (in_lds)
            for (int q = tid; q < csz; q += kLwThreads) sh_cpt[q] = a.cpt[coff + q];
        __syncthreads();
        if ((t & 1) == 0 && draws) {  // fresh Philox block for positions t and t+1
#pragma unroll
            for (int r = 0; r < kLwPerThread; ++r) {
                const uint64_t s = a.sample_base + local0 + uint64_t(r) * kLwThreads;
                philox4x32_10(uint32_t(s), uint32_t(s >> 32), uint32_t(t >> 1), 0u, key0, key1, rnd[r]);
            }
        } else if ((t & 1) == 1 && draws) {
            // the even position may have been a clamped evidence node (no block drawn yet)
            const int vprev = a.topo[t - 1];
            if (!reject && a.ev_state[vprev] >= 0) {
#pragma unroll
                for (int r = 0; r < kLwPerThread; ++r) {
                    const uint64_t s = a.sample_base + local0 + uint64_t(r) * kLwThreads;
                    philox4x32_10(uint32_t(s), uint32_t(s >> 32), uint32_t(t >> 1), 0u, key0, key1, rnd[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kLwPerThread; ++r) {
            const uint64_t col = local0 + uint64_t(r) * kLwThreads;
            int64_t row = 0;
            for (int e = e0; e < e1; ++e) {
                const int p = a.in_idx[e];
                row = row * a.k[p] + a.states[uint64_t(p) * a.batch + col];
            }
            const double* rowp = in_lds ? (sh_cpt + row * kv) : (a.cpt + coff + row * kv);
            int st;
            if (ev >= 0 && !reject) {
                w[r] *= rowp[ev];
                st = ev;
            } else {
                const uint32_t lo = (t & 1) ? rnd[r][2] : rnd[r][0], hi = (t & 1) ? rnd[r][3] : rnd[r][1];
                const uint64_t x = (uint64_t(hi) << 32) | lo;
                const double u = double(x >> 11) * (1.0 / 9007199254740992.0);
                st = kv - 1;
                bool found = false;
                double total = 0.0;
                for (int i = 0; i < kv; ++i) {
                    const double old_total = total;
                    total += rowp[i];
                    if (!found && old_total <= u && u < total) { st = i; found = true; }
                }
                if (ev >= 0 && st != ev) w[r] = 0.0;  // rejected (rejection_sampling.hpp:70-84)
            }
            a.states[uint64_t(v) * a.batch + col] = uint8_t(st);
        }
    }
#pragma unroll
    for (int r = 0; r < kLwPerThread; ++r) a.weights[local0 + uint64_t(r) * kLwThreads] = w[r];
}

// hist[v][state_v] += w for the first n_valid samples of the batch, w being FINAL (every evidence
// node visited).  Same sample-to-thread mapping as the sampling kernel.
__global__ __launch_bounds__(kLwThreads) void lw_hist_kernel(LwArgs a) {
    __shared__ double sh_hist[256];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint64_t local0 = uint64_t(blockIdx.x) * kLwBlockSamples + tid;
    double w[kLwPerThread];
    bool valid[kLwPerThread];
#pragma unroll
    for (int r = 0; r < kLwPerThread; ++r) {
        valid[r] = (local0 + uint64_t(r) * kLwThreads) < a.n_valid;
        w[r] = valid[r] ? a.weights[local0 + uint64_t(r) * kLwThreads] : 0.0;
    }

    for (int v = 0; v < a.n; ++v) {
        const int kv = a.k[v];
        __syncthreads();
        if (tid < kv) sh_hist[tid] = 0.0;
        __syncthreads();
        double acc8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc8[i] = 0.0;
#pragma unroll
        for (int r = 0; r < kLwPerThread; ++r) {
            if (!valid[r]) continue;
            const int st = a.states[uint64_t(v) * a.batch + local0 + uint64_t(r) * kLwThreads];
            if (kv <= 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc8[i] += (i == st) ? w[r] : 0.0;
            } else {
                atomicAdd(&sh_hist[st], w[r]);
            }
        }
        if (kv <= 8) {
            for (int i = 0; i < kv; ++i) {
                double x = 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) x = (q == i) ? acc8[q] : x;
                x = wave_sum(x);
                if (lane == 0 && x != 0.0) atomicAdd(&sh_hist[i], x);
            }
        }
        __syncthreads();
        if (tid < kv) {
            const double x = sh_hist[tid];
            if (x != 0.0) atomicAdd(&a.hist[a.node_off[v] + tid], x);
        }
    }
}

int launch_lw_sample(const LwArgs& a, int blocks, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    hipLaunchKernelGGL(lw_sample_kernel, dim3(blocks), dim3(kLwThreads), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}
int launch_lw_hist(const LwArgs& a, int blocks, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    hipLaunchKernelGGL(lw_hist_kernel, dim3(blocks), dim3(kLwThreads), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

}  // namespace bnmi

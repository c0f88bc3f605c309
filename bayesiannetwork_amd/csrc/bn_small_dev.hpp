// bn_small_dev.hpp -- device helpers shared by bn_small.hip and bn_mid.hip: wave-level reductions without LDS traffic.
#pragma once

#include <hip/hip_runtime.h>

namespace bnmi {

__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// max over the wave's lanes with DPP moves (no LDS traffic: __shfl_xor is a ds_bpermute per step, and 64 lanes of one
// LDS atomic on the same word cost ~2.4 us): row_shr 1, 2, 4, 8 leave each row's maximum in its lane 15, row_bcast:15 /
// row_bcast:31 carry it on to lane 63.  ROW = true stops after the rows (the maximum of lanes 0..15 in lane 15).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax_step(unsigned x) {
    const unsigned o = unsigned(__builtin_amdgcn_update_dpp(0, int(x), CTRL, ROW_MASK, 0xf, true));
    return o > x ? o : x;
}
template <bool ROW = false>
__device__ __forceinline__ unsigned wave_umax32_dpp(unsigned x) {
    x = dpp_umax_step<0x111, 0xf>(x);
    x = dpp_umax_step<0x112, 0xf>(x);
    x = dpp_umax_step<0x114, 0xf>(x);
    x = dpp_umax_step<0x118, 0xf>(x);
    if (ROW) return unsigned(__builtin_amdgcn_readlane(int(x), 15));
    x = dpp_umax_step<0x142, 0xa>(x);
    x = dpp_umax_step<0x143, 0xc>(x);
    return unsigned(__builtin_amdgcn_readlane(int(x), 63));
}
// ... of 64-bit words: the high halves first, then the low halves of the lanes that hold the largest high half
template <bool ROW = false>
__device__ __forceinline__ unsigned long long wave_umax64_dpp(unsigned long long v) {
    const unsigned hi = unsigned(v >> 32), lo = unsigned(v);
    const unsigned hm = wave_umax32_dpp<ROW>(hi);
    const unsigned lm = wave_umax32_dpp<ROW>(hi == hm ? lo : 0u);
    return (unsigned long long)hm << 32 | lm;
}

__device__ __forceinline__ int wave_imax(int x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int o = __shfl_xor(x, off, 64);
        x = o > x ? o : x;
    }
    return __builtin_amdgcn_readfirstlane(x);
}

}  // namespace bnmi

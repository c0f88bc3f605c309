// bn_lw_kernels.hip -- likelihood weighting, reference bayesian/inference/likelihood_weighting.hpp.
//
// Sampling (lw_sample_kernel): one thread draws kLwPerThread = 4 CONSECUTIVE ancestral samples, so
// the four states of a node are one dword of the [node][sample] byte matrix (a wave moves 256
// contiguous bytes per load/store).  Waves never synchronise: every wave walks the topological
// order on its own, the per-position metadata (LwStep, parent list) is wave-uniform and comes
// through scalar loads one position ahead, CPT rows are gathered straight from global memory
// (a node's table is at most a few KiB and shared by every wave: L1/L2 hits).  Per node and sample
// (weighted_sample, :122-173):
//   row   = parent assignment, first parent most significant, from the state matrix
//   evidence node  : w *= cpt[row][ev], state = ev                         (:148-153)
//   otherwise      : state = first i with cum_{i-1} <= u < cum_i, else k-1   (:154-158, :177-193)
// (rejection / logic sampling, reference rejection_sampling.hpp:33-167, is the same walk with the
// evidence nodes sampled like any other and w = 1 if every one of them came out as observed, else 0).
// Uniforms: every sample owns a xoshiro128++ stream seeded by one Philox4x32-10 block keyed by
// (seed, global sample id) and advanced by ONE step per TWO topological positions: the top half of
// the ++ output is the top 16 bits of the even position's 53-bit uniform, the bottom half those of
// the odd position after it; the 37 bits below come from the ** scrambler of words of the state the
// step left behind and are looked at only when the top 16 tie with a threshold -- see
// oracle/lw_oracle.c for the exact mapping, which this kernel reproduces bit for bit, so sampled
// states are identical.  (Philox for every draw, 20 quarter-rate 32x32->64 multiplies per pair of
// positions, made the kernel VALU-bound at 62 % Philox; two xoshiro steps per uniform were 20
// full-rate ops; one step per position, round 4, was ten -- a fifth of the vector instructions of a
// kernel bound by vector issue, for 32 deciding bits where 16 decide all but 3 draws in 2^16.)
//
// Histogram (lw_hist_kernel, :45-49, run once the weights are FINAL): lane = node, each lane
// streams its own row of the state matrix 16 samples at a time and adds the (wave-uniform) weights
// into per-state accumulators kept in registers -- no cross-lane reduction; one fp64 atomicAdd per
// (sample range, node, state) at the end.  Nodes with more than 8 states use lw_hist_wide_kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "bn_lw.hpp"

namespace bnmi {

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = uint64_t(0xD2511F53u) * c0, p1 = uint64_t(0xCD9E8D57u) * c2;
        const uint32_t n0 = uint32_t(p1 >> 32) ^ c1 ^ k0, n2 = uint32_t(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = uint32_t(p1); c2 = n2; c3 = uint32_t(p0);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// xoshiro128++ 1.0 (Blackman & Vigna): one 32-bit output, state in x/y/z/w
__device__ __forceinline__ uint32_t xoshiro_next(uint4& g) {
    const uint32_t sum = g.x + g.w;
    const uint32_t result = ((sum << 7) | (sum >> 25)) + g.x;
    const uint32_t t = g.y << 9;
    g.z ^= g.x; g.w ^= g.y; g.y ^= g.z; g.x ^= g.w;
    g.z ^= t;
    g.w = (g.w << 11) | (g.w >> 21);
    return result;
}

// the ** scrambler of one state word
__device__ __forceinline__ uint32_t xoshiro_ss(uint32_t w) {
    const uint32_t x = w * 5u;
    return ((x << 7) | (x >> 25)) * 9u;
}
// Position parity PAR (0: the position whose step this was, 1: the one after it).  h16 = the position's half of the step's output;
// the 53-bit integer of its uniform, u = U * 2^-53, takes 37 more bits from the state as the step left it (oracle/lw_oracle.c)
template <int PAR>
__device__ __forceinline__ uint32_t half_of(uint32_t out) { return PAR == 0 ? out >> 16 : out & 0xffffu; }
template <int PAR>
__device__ __forceinline__ unsigned long long unit53(uint32_t h16, const uint4& g_after) {
    const unsigned long long low = PAR == 0 ? (uint64_t(xoshiro_ss(g_after.y)) << 5) | (xoshiro_ss(g_after.z) >> 27)
                                            : (uint64_t(xoshiro_ss(g_after.w)) << 5) | (xoshiro_ss(g_after.x) >> 27);
    return (uint64_t(h16) << 37) | low;
}

// States of the S samples of one thread at one node: first i with cum_{i-1} <= u < cum_i, else
// KV-1 (:177-193).  All rows are loaded before the first compare so the S gathers overlap.
// The uniforms are drawn between issuing the row gathers and using them (latency cover).
// KV = 2, 3, 4: the running totals of a row were added up ONCE, on the host, in the reference's left-to-right order, and are kept
// as 64-bit integer thresholds T_i = ceil(total_i * 2^53).  A uniform is u = U * 2^-53 with the 53-bit integer U the two generator
// words give, so "u >= total_i" (the comparison the per-draw additions fed) is EXACTLY "U >= T_i": total_i * 2^53 is the same double
// scaled by a power of two, and an integer is >= a real number iff it is >= its ceiling.  Same states bit for bit, without k - 1 fp64
// additions per draw and without the integer -> double conversion of the uniform.
// The top 32 bits of U are the generator's output as it stands, so the draw is settled by 32-bit compares against the top
// halves T_i >> 21 -- ONE 16-byte gather per draw (the sampler is bound by its row gathers as much as by its vector ALU: 32-byte
// rows of doubles were two gather instructions and twice the cache lines) -- unless a top half ties (3 x 2^-32 per draw): then
// the wave repeats that draw against the full thresholds.
typedef unsigned lw_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned lw_u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned lw_u32x2 __attribute__((ext_vector_type(2)));
template <int KV>
__device__ __forceinline__ void load_top(__amdgpu_buffer_rsrc_t rs, uint32_t row, uint32_t (&t)[3]) {
    t[0] = t[1] = t[2] = 0xffffffffu;
    if constexpr (KV == 4) {
        const lw_u32x4 q = __builtin_bit_cast(lw_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, int(row * 16u), 0, 0));
        t[0] = q.x; t[1] = q.y; t[2] = q.z;
    } else if constexpr (KV == 3) {
        const lw_u32x3 q = __builtin_bit_cast(lw_u32x3, __builtin_amdgcn_raw_buffer_load_b96(rs, int(row * 12u), 0, 0));
        t[0] = q.x; t[1] = q.y;
    } else {
        const lw_u32x2 q = __builtin_bit_cast(lw_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, int(row * 8u), 0, 0));
        t[0] = q.x;
    }
}
// `out`: the ++ outputs of the step the position belongs to (taken by the caller at the even position); PAR: the position's parity.
// A draw is settled by its 16 bits against the top 16 of the thresholds (the top halves T_i >> 21 of d_thr32, shifted) unless
// they tie (3 x 2^-16 per draw): then the wave repeats the position's draws against the full thresholds.
template <int KV, int S, int PAR>
__device__ __forceinline__ void pick_states(const double* __restrict__ base, const unsigned long long* __restrict__ tbase,
                                            const uint32_t* __restrict__ tbase32, const uint32_t (&row)[S], const uint32_t (&out)[S],
                                            const uint4 (&rng)[S], int kv, int (&st)[S]) {
    if (KV > 0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(tbase32), 0, 0x7fffffff, 0x00020000);
        uint32_t t[S][3];
#pragma unroll
        for (int r = 0; r < S; ++r) load_top<KV>(rs, row[r], t[r]);
        bool tie = false;
#pragma unroll
        for (int r = 0; r < S; ++r) {
            const uint32_t h = half_of<PAR>(out[r]);
            int c = 0;
#pragma unroll
            for (int i = 0; i < KV - 1; ++i) {
                c += (h > (t[r][i] >> 16)) ? 1 : 0;
                tie = tie || h == (t[r][i] >> 16);
            }
            st[r] = c;
        }
        if (__any(tie)) {   // (wave-uniform branch: the four draws again, against the full thresholds)
#pragma unroll
            for (int r = 0; r < S; ++r) {
                const unsigned long long U = unit53<PAR>(half_of<PAR>(out[r]), rng[r]);
                const unsigned long long* rowp = tbase + uint64_t(row[r]) * KV;
                int c = 0;
#pragma unroll
                for (int i = 0; i < KV - 1; ++i) c += (U >= rowp[i]) ? 1 : 0;
                st[r] = c;
            }
        }
    } else {
        double u[S];
        bool found[S];
        double total[S];
        const double* rowp[S];
#pragma unroll
        for (int r = 0; r < S; ++r) {
            found[r] = false; total[r] = 0.0; st[r] = kv - 1; rowp[r] = base + uint64_t(row[r]) * kv;
            u[r] = double(unit53<PAR>(half_of<PAR>(out[r]), rng[r])) * (1.0 / 9007199254740992.0);
        }
        for (int i = 0; i < kv; ++i) {
            double x[S];
#pragma unroll
            for (int r = 0; r < S; ++r) x[r] = rowp[r][i];
#pragma unroll
            for (int r = 0; r < S; ++r) {
                const double old_total = total[r];
                total[r] += x[r];
                if (!found[r] && old_total <= u[r] && u[r] < total[r]) { st[r] = i; found[r] = true; }
            }
        }
    }
}

// One draw against the three 16-bit thresholds of its row (ex = e0 | e1 << 16, ey = e2 | 0xffff << 16; h = the position's half of
// `out`): count = how many of them h exceeds, nearest = min_i (e_i - h) mod 2^32, zero exactly when h ties with one.  The subtraction's
// borrow IS "h > e_i": subtract-with-carry-out (operands picked out of their registers by SDWA selects, no extraction) followed by
// add-with-carry.  Hand-written because the compiler, given the same arithmetic, parks every borrow in a scalar register pair and
// extracts every half with an instruction of its own.  An SDWA instruction that writes VCC needs two wait states before a vector
// instruction reads it (the compiler's own code keeps that distance): s_nop 1.
template <int PAR>
__device__ __forceinline__ void count_and_tie(uint32_t ex, uint32_t ey, uint32_t out, uint32_t& count, uint32_t& nearest) {
    uint32_t d0, d1, d2, c;
    if (PAR == 0) {
        asm("v_sub_co_u32_sdwa %0, vcc, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32_e64 %3, 0, 1, vcc\n\t"
            "v_sub_co_u32_sdwa %1, vcc, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
            "v_sub_co_u32_sdwa %2, vcc, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc"
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(c)
            : "v"(ex), "v"(ey), "v"(out)
            : "vcc");
    } else {
        asm("v_sub_co_u32_sdwa %0, vcc, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32_e64 %3, 0, 1, vcc\n\t"
            "v_sub_co_u32_sdwa %1, vcc, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
            "v_sub_co_u32_sdwa %2, vcc, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc"
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(c)
            : "v"(ex), "v"(ey), "v"(out)
            : "vcc");
    }
    count = c;
    nearest = min(min(d0, d1), d2);
}

// The same selection for nodes with <= 256 CPT rows (kLwStepPacked), thresholds read from the WAVE'S OWN COPY of the node's table
// in LDS.  What bounded the sampler (round 4, TCP counters: the vector L1 busy 97 % of the kernel, 160 tag look-ups per wave
// and position, 110 of them the four row gathers -- 64 lanes x 16 bytes scattered over a 4 KB table are ~28 cache lines per
// instruction) was the gather, not arithmetic and not latency.  So the table comes in with COALESCED loads (the whole table, 16
// bytes per lane and instruction), goes to LDS, and the per-sample rows are ds_reads.  To halve both the copy and the LDS
// traffic the copy holds the top 16 bits of each threshold (T >> 37): one 8-byte row {t0 | t1 << 16, t2 | pad}.  A draw whose
// top 16 bits equal one of its row's entries is undecided (3 x 2^-16 per draw, ~1 % of a wave's positions): the wave then
// repeats the position's draws against the full 64-bit thresholds in memory.  States are the same bit for bit either way.
template <int KV, int S, int PAR>
__device__ __forceinline__ void pick_states16(const uint2* tab, const unsigned long long* __restrict__ tbase, const uint32_t (&row)[S],
                                              const uint32_t (&out)[S], const uint4 (&rng)[S], int (&st)[S]) {
    uint2 t[S];
#pragma unroll
    for (int r = 0; r < S; ++r) t[r] = tab[row[r]];
    bool tie = false;
#pragma unroll
    for (int r = 0; r < S; ++r) {
        const uint32_t h = half_of<PAR>(out[r]);
        const uint32_t e[3] = {t[r].x & 0xffffu, t[r].x >> 16, t[r].y & 0xffffu};
        int c = 0;
#pragma unroll
        for (int i = 0; i < KV - 1; ++i) {
            c += (h > e[i]) ? 1 : 0;
            tie = tie || h == e[i];
        }
        st[r] = c;
    }
    if (__any(tie)) {
#pragma unroll
        for (int r = 0; r < S; ++r) {
            const unsigned long long U = unit53<PAR>(half_of<PAR>(out[r]), rng[r]);
            const unsigned long long* rowp = tbase + uint64_t(row[r]) * KV;
            int c = 0;
#pragma unroll
            for (int i = 0; i < KV - 1; ++i) c += (U >= rowp[i]) ? 1 : 0;
            st[r] = c;
        }
    }
}

typedef uint8_t __attribute__((address_space(1))) * lw_global_bytes;   // (global address space spelled out: a pointer made from an integer is a flat one otherwise)
typedef uint32_t __attribute__((address_space(1))) * lw_global_u32;

struct LwStepWords {  // LwStep as two 16-byte words
    uint4 a, b;
};

#ifndef BN_LW_WAVES
#define BN_LW_WAVES 5
#endif
template <bool ROWS24, bool INLINE, bool REJECT>
__global__ __launch_bounds__(kLwThreads) __attribute__((amdgpu_waves_per_eu(BN_LW_WAVES - 1, BN_LW_WAVES))) void lw_sample_kernel(
    const LwStepWords* __restrict__ steps, const uint4* __restrict__ parents, const int32_t* __restrict__ ev_topo,
    const double* __restrict__ cpt, const unsigned long long* __restrict__ thr, const uint32_t* __restrict__ thr32,
    const uint4* __restrict__ thr16, uint8_t* states,
    double* __restrict__ weights, int32_t n,
    uint64_t batch, uint64_t sample_base, uint64_t seed) {
    constexpr int S = kLwPerThread;
    static_assert(S == 4, "one dword of states per thread and node");
    static_assert(sizeof(LwStep) == 32 && sizeof(LwParent) == 8, "descriptor layout");
    const uint32_t col32 = (blockIdx.x * kLwThreads + threadIdx.x) * S;  // first sample of this thread: its byte offset in a row of the state matrix (batch < 2^31, bn_lw.cpp)
    const uint64_t col = col32;
    constexpr bool reject = REJECT;  // logic (rejection) sampling: evidence nodes are drawn like any other, w in {0, 1}
    double w[S];
    uint4 rng[S];
    const uint32_t key0 = uint32_t(seed), key1 = uint32_t(seed >> 32);
#pragma unroll
    for (int r = 0; r < S; ++r) {
        w[r] = 1.0;  // :124
        const uint64_t s = sample_base + col + r;
        rng[r] = philox4x32_10(uint32_t(s), uint32_t(s >> 32), 0u, 0u, key0, key1);
        if ((rng[r].x | rng[r].y | rng[r].z | rng[r].w) == 0) rng[r].x = 1;
    }

    // A row of the state matrix is addressed as {scalar base of the node's row} + {this thread's 32-bit byte offset}: the base
    // is scalar arithmetic and the load takes it from SGPRs -- no vector arithmetic per parent (node * batch + col as a 64-bit
    // vector mad was a quarter-rate instruction per parent and position).
    // (the halves of the base go through an empty asm with an SGPR constraint: left alone, the compiler re-associates the sum into
    // one 64-bit vector multiply-add per row)
    auto row_of = [&](uint32_t node) {
        const uint64_t b = reinterpret_cast<uint64_t>(states) + uint64_t(node) * batch;
        uint32_t lo = uint32_t(b), hi = uint32_t(b >> 32);
        asm volatile("" : "+s"(lo), "+s"(hi));
        return reinterpret_cast<lw_global_bytes>((uint64_t(hi) << 32) | lo);
    };
    // (... and the 32-bit offset through one with a VGPR constraint in every iteration: hoisted out of the loop as a 64-bit value, it is no
    // longer recognised as the zero-extended offset of the SGPR-base addressing form)
    uint32_t c32 = col32;
    auto off32 = [&]() { return c32; };
    auto ld = [&](uint32_t node) { return *reinterpret_cast<lw_global_u32>(row_of(node) + off32()); };
    __shared__ uint4 lw_tab[kLwThreads / 64][128];   // per wave: the 16-bit thresholds of the node it is at (<= 256 rows x 8 bytes)
    const uint32_t lane = threadIdx.x & 63u;
    uint4* const my_tab = lw_tab[threadIdx.x >> 6];

    // the first four parents of a position: one branch-free load group per parent count
    auto request = [&](const uint4& sdw, const uint4& pw, uint32_t (&w4)[4]) {
        switch (int((sdw.w >> 24) & 0x1fu) < 4 ? int((sdw.w >> 24) & 0x1fu) : 4) {
            case 0: break;
            case 1: w4[0] = ld(pw.x & 0xffffffu); break;
            case 2: w4[0] = ld(pw.x & 0xffffffu); w4[1] = ld(pw.y & 0xffffffu); break;
            case 3: w4[0] = ld(pw.x & 0xffffffu); w4[1] = ld(pw.y & 0xffffffu); w4[2] = ld(pw.z & 0xffffffu); break;
            default:
                w4[0] = ld(pw.x & 0xffffffu); w4[1] = ld(pw.y & 0xffffffu); w4[2] = ld(pw.z & 0xffffffu); w4[3] = ld(pw.w & 0xffffffu);
                break;
        }
    };

    // (Requesting the parents' states of position t + 1 before position t is worked on -- wherever t's node is not among them --
    // was built and measured SLOWER, 32.1 vs 28.4 ms per 2 M samples: the kernel was bound by the vector L1's look-up rate, and
    // more loads in flight only queue there.)  Descriptors come through scalar loads one position ahead (both arrays have
    // spare entries at the end).
    LwStepWords nxt = steps[0];
    int ev_nxt = ev_topo[0];
    uint32_t out[S];   // the ++ outputs of the last even position's step: their top halves decide it, the bottom halves the odd position after it
#pragma unroll
    for (int r = 0; r < S; ++r) out[r] = 0;
    auto position = [&](int t, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        asm volatile("" : "+v"(c32));
        const uint4 sd = nxt.a;  // LwStep: coff_lo, v, par_off, coff_hi | kv << 16 | (m | flags) << 24
        const uint4 pin = nxt.b;  // parents 0..3 inline: node | arity << 24
        const int ev = ev_nxt;
        nxt = steps[t + 1];
        ev_nxt = ev_topo[t + 1];
        const int kv = int((sd.w >> 16) & 0xffu), m = int((sd.w >> 24) & 0x1fu);
        const bool byte_rows = INLINE && (sd.w & (uint32_t(kLwStepPacked) << 24)) != 0;
        const bool draws = reject || ev < 0;  // logic sampling draws evidence nodes too
        const bool staged = byte_rows && draws && kv <= 4;
        uint4 q0, q1;   // (only looked at when staged)
        if (staged) {   // the node's table of 16-bit thresholds: two coalesced 1 KB pieces (sd.z = its place in thr16, in rows)
            const uint4* src = thr16 + (sd.z >> 1);
            q0 = src[lane];
            q1 = src[64 + lane];
        }
        uint32_t wd[4] = {0, 0, 0, 0};
        if (INLINE) request(sd, pin, wd);
        if (PAR == 0) {   // the step of this position and the next, evidence node or not: u(s, t) does not depend on the evidence
#pragma unroll
            for (int r = 0; r < S; ++r) out[r] = xoshiro_next(rng[r]);
        }
        uint32_t row[S];
#pragma unroll
        for (int r = 0; r < S; ++r) row[r] = 0;
        if (INLINE) {
            const uint32_t pk[4] = {pin.x >> 24, pin.y >> 24, pin.z >> 24, pin.w >> 24};   // arities; 0 = no such parent (its word is 0)
            if (byte_rows) {
                // <= 4 parents and <= 256 rows: the four samples' row numbers as the four BYTES of one register.  Every partial
                // row number is below the product of the arities so far (<= 256), so a byte never carries into its neighbour:
                // one multiply-add (arities that are powers of two: one shift-add) per parent for all four samples, instead
                // of an extract and a multiply-add per parent AND sample.
                uint32_t rp = 0;
                if (sd.w & (uint32_t(kLwStepPow2) << 24)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) rp = (rp << (pk[j] ? __builtin_ctz(pk[j]) : 0)) + wd[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) rp = rp * (pk[j] ? pk[j] : 1u) + wd[j];
                }
#pragma unroll
                for (int r = 0; r < S; ++r) row[r] = (rp >> (8 * r)) & 0xffu;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < m) {
#pragma unroll
                        for (int r = 0; r < S; ++r)
                            row[r] = (ROWS24 ? __umul24(row[r], pk[j]) : row[r] * pk[j]) + ((wd[j] >> (8 * r)) & 0xffu);
                    }
                }
            }
        }
        for (int j0 = INLINE ? 4 : 0; j0 < m; j0 += 4) {  // parent lists are padded to pairs: two LwParent per uint4
            const uint4 pa = parents[(sd.z + j0) >> 1];
            const uint4 pb = (j0 + 2 < m) ? parents[((sd.z + j0) >> 1) + 1] : make_uint4(0, 1, 0, 1);
            uint32_t we[4] = {ld(pa.x), 0, 0, 0};
            if (j0 + 1 < m) we[1] = ld(pa.z);
            if (j0 + 2 < m) we[2] = ld(pb.x);
            if (j0 + 3 < m) we[3] = ld(pb.z);
            const uint32_t kk[4] = {pa.y, (j0 + 1 < m) ? pa.w : 1u, (j0 + 2 < m) ? pb.y : 1u, (j0 + 3 < m) ? pb.w : 1u};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < S; ++r)
                    row[r] = (ROWS24 ? __umul24(row[r], kk[q]) : row[r] * kk[q]) + ((we[q] >> (8 * r)) & 0xffu);
        }
        const uint64_t coff = (uint64_t(sd.w & 0xffffu) << 32) | sd.x;
        const double* base = cpt + coff;
        const unsigned long long* tbase = thr + coff;
        const uint32_t* tbase32 = thr32 + coff;
        uint32_t packed = 0;
        if (!draws) {
            double x[S];
#pragma unroll
            for (int r = 0; r < S; ++r) x[r] = base[uint64_t(row[r]) * kv + ev];
#pragma unroll
            for (int r = 0; r < S; ++r) w[r] *= x[r];
            packed = uint32_t(ev) * 0x01010101u;
        } else {
            int st[S];
            if (staged) {
                my_tab[lane] = q0;
                my_tab[64 + lane] = q1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint2* tab = reinterpret_cast<const uint2*>(my_tab);
                switch (kv) {
                    case 2: pick_states16<2, S, PAR>(tab, tbase, row, out, rng, st); break;
                    case 3: pick_states16<3, S, PAR>(tab, tbase, row, out, rng, st); break;
                    default: pick_states16<4, S, PAR>(tab, tbase, row, out, rng, st); break;
                }
            } else switch (kv) {
                case 2: pick_states<2, S, PAR>(base, tbase, tbase32, row, out, rng, 2, st); break;
                case 3: pick_states<3, S, PAR>(base, tbase, tbase32, row, out, rng, 3, st); break;
                case 4: pick_states<4, S, PAR>(base, tbase, tbase32, row, out, rng, 4, st); break;
                default: pick_states<0, S, PAR>(base, tbase, tbase32, row, out, rng, kv, st); break;
            }
#pragma unroll
            for (int r = 0; r < S; ++r) {
                if (REJECT && ev >= 0 && st[r] != ev) w[r] = 0.0;  // rejected (rejection_sampling.hpp:70-84)
                packed |= uint32_t(st[r]) << (8 * r);
            }
        }
        *reinterpret_cast<lw_global_u32>(row_of(sd.y) + off32()) = packed;
    };
    int t = 0;
    for (; t + 1 < n; t += 2) {
        position(t, std::integral_constant<int, 0>());
        position(t + 1, std::integral_constant<int, 1>());
    }
    if (t < n) position(t, std::integral_constant<int, 0>());
#pragma unroll
    for (int r = 0; r < S; ++r) weights[col + r] = w[r];
}

// Networks in which EVERY node has <= 4 parents, <= 256 CPT rows and <= 4 states (LwState::small; n < 2^24): the same walk without
// the generic kernel's case distinctions.  Round 4's counters, after the gathers had been taken off the vector L1: the kernel
// is bound by instruction ISSUE -- per wave and position 128 vector + 108 scalar instructions + 21 branches, a wave issuing one
// instruction per turn.  Here a position is straight-line code with two wave-uniform branches (evidence node; a tie):
//   * four parent loads always -- a missing parent points at the all-zero row behind the state matrix, with arity 1;
//   * the row numbers of the thread's four samples are the four BYTES of one register (three shift-adds, or multiply-adds, for
//     all four samples; every partial row number is below 256, so a byte never carries into its neighbour);
//   * the node's 16-bit thresholds (8 bytes per row; the copy stops where the table ends) go through the wave's LDS slice;
//     always three compares per draw -- a node with fewer states has 0xffff in the unused places, which no draw exceeds;
//   * the stream steps at the even positions only: the loop body is instantiated for both parities and runs them in turn.
// POW2: every arity of the network is a power of two (LwSmallStep::shape holds log2 of the arities instead of the arities).
// The descriptor (LwSmallStep, 64 bytes, one scalar load a position ahead) holds the rows' BYTE OFFSETS in the state matrix ready:
// a row's base is one 64-bit scalar add (node x stride as scalar multiplies was 8 scalar instructions per row, 40 per position).
#ifndef BN_LW_SMALL_WAVES
#define BN_LW_SMALL_WAVES 7
#endif
struct LwSmallWords {  // LwSmallStep as four 16-byte words
    uint4 a, b, c, d;
};
template <bool POW2, bool REJECT>
__global__ __launch_bounds__(kLwThreads) __attribute__((amdgpu_waves_per_eu(BN_LW_SMALL_WAVES - (REJECT ? 1 : 0), BN_LW_SMALL_WAVES))) void lw_sample_small_kernel(
    const LwSmallWords* __restrict__ steps, const int32_t* __restrict__ ev_topo, const double* __restrict__ cpt,
    const unsigned long long* __restrict__ thr, const uint4* __restrict__ thr16, uint8_t* states, double* __restrict__ weights, int32_t n,
    uint64_t sample_base, uint64_t seed) {
    constexpr int S = kLwPerThread;
    static_assert(S == 4, "one dword of states per thread and node");
    static_assert(sizeof(LwSmallStep) == 64, "descriptor layout");
    const uint32_t col32 = (blockIdx.x * kLwThreads + threadIdx.x) * S;   // first sample of this thread
    const uint32_t colb = blockIdx.x * kLwThreads + threadIdx.x;          // ... and its byte in a row of the state matrix: four samples per byte
    double w[S];
    uint4 rng[S];
    const uint32_t key0 = uint32_t(seed), key1 = uint32_t(seed >> 32);
#pragma unroll
    for (int r = 0; r < S; ++r) {
        w[r] = 1.0;  // :124
        const uint64_t s = sample_base + col32 + r;
        rng[r] = philox4x32_10(uint32_t(s), uint32_t(s >> 32), 0u, 0u, key0, key1);
        if ((rng[r].x | rng[r].y | rng[r].z | rng[r].w) == 0) rng[r].x = 1;
    }
    // a row of the state matrix = {scalar base: the row's address, which the descriptor holds ready} + {this thread's 32-bit offset};
    // see lw_sample_kernel for the two empty asm statements
    auto row_at = [&](uint32_t addr_lo, uint32_t addr_hi) {
        uint32_t lo = addr_lo, hi = addr_hi;
        asm volatile("" : "+s"(lo), "+s"(hi));
        return reinterpret_cast<lw_global_bytes>((uint64_t(hi) << 32) | lo);
    };
    uint32_t c32 = colb;
    // THE STATE MATRIX HOLDS TWO BITS PER STATE (every arity of these networks is <= 4): a thread's four samples of a node are one
    // byte.  Round 5: with one byte per state the sampler wrote 10 KB per sample, read ~20 KB of parents' rows back (a wave's rows of
    // the last 64 nodes are 16 KB, times 8 192 waves: beyond the L2) and the histogram pass read the 10 KB again -- 43 KB per sample,
    // 4.3 TB/s at 1.0e8 samples/s, on a box whose device-to-device copy reaches 5.2: every on-chip change (vector, scalar, LDS,
    // vector-memory instruction counts) left the sampling kernel's time where it was.  A parent's byte is spread to the four bytes
    // of a register (three vector instructions) so that the row numbers are still formed for four samples at once.
    typedef uint8_t __attribute__((address_space(1))) * lw_global_u8;
    auto spread = [](uint32_t b) {   // bits 2r .. 2r + 1 -> byte r
        uint32_t x = b | (b << 6);
        x = x | (x << 12);
        return x & 0x03030303u;
    };
    __shared__ uint4 lw_tab[kLwThreads / 64][128];
    const uint32_t lane = threadIdx.x & 63u;
    uint4* const my_tab = lw_tab[threadIdx.x >> 6];
    const uint2* const tab = reinterpret_cast<const uint2*>(my_tab);

    // What a position reads from memory, and only what it needs (the descriptor says how many parents the node has and how long its
    // table is; the branches on them are scalar):
    //   * the parents' states of this thread's four samples, one byte each -- m loads, not always four;
    //   * this lane's 16 bytes of the node's table of 16-bit thresholds, through a buffer descriptor that ENDS with the table (lanes
    //     beyond it get zeros and request nothing); the second kilobyte only for tables that have one (256 rows);
    //   * a ROOT has one row: its three thresholds are wave-uniform and come through a scalar load -- no table copy, no LDS.
    // (Round 4 copied 2 KB per position whatever the table's size -- ~10 TB/s of L2 traffic on config 5 -- and always loaded four
    // parents, the missing ones from an all-zero row.)
    struct Bytes { uint32_t b0, b1, b2, b3; };
    struct Tabs { uint4 q0, q1; };
    // ALWAYS the same loads per position, so that the number in flight is known when the code is compiled: a wave waits for "all but
    // the six issued last" (s_waitcnt vmcnt(6)), i.e. for its own position's loads and not for the later positions'.  With a number of
    // loads that depends on the node the compiler has to wait for all of them, and the pipeline overlaps nothing (measured: 68.3 ms either
    // way).  A missing parent is the all-zero row n; a table load beyond the table's end requests nothing.
    auto fetch_bytes = [&](const LwSmallWords& d, Bytes& p) {
        p.b0 = *reinterpret_cast<lw_global_u8>(row_at(d.a.x, d.a.y) + c32);
        p.b1 = *reinterpret_cast<lw_global_u8>(row_at(d.a.z, d.a.w) + c32);
        p.b2 = *reinterpret_cast<lw_global_u8>(row_at(d.b.x, d.b.y) + c32);
        p.b3 = *reinterpret_cast<lw_global_u8>(row_at(d.b.z, d.b.w) + c32);
    };
    auto fetch_tabs = [&](const LwSmallWords& d, Tabs& p) {
        const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<void*>((uint64_t(d.c.w) << 32) | d.c.z), 0, int(d.d.x), 0x00020000);   // (LwSmallStep::tab: base, bytes)
        p.q0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(trs, int(lane * 16u), 0, 0));
        p.q1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(trs, int(lane * 16u + 1024u), 0, 0));
    };
    auto patch = [&](Bytes& p, uint32_t j, uint32_t byte) {
        if (j == 0) p.b0 = byte;
        else if (j == 1) p.b1 = byte;
        else if (j == 2) p.b2 = byte;
        else p.b3 = byte;
    };
    // SOFTWARE PIPELINE (round 5).  With two bits per state the kernel is bound by LATENCY -- 4 / 6 / 8 waves per SIMD draw 73 / 93 /
    // 105 M samples/s, and a wave spends ~3 000 cycles per position, most of them waiting for its parents' bytes (a wave's rows of the
    // last 64 nodes, times 8 192 waves, do not fit the L2: the reads come back from the Infinity Cache).  So a position's loads are
    // issued AHEAD: its parents' bytes two positions before it is worked on, its table one position before.  Where a parent is the
    // node of the position before or of the one before that (LwSmallStep::shape bits 7 and 22: ~6 % of config 5's positions each) the
    // byte loaded ahead is stale: it is replaced, behind that position's store, by the byte just stored.  (Rounds 4 and 5 measured a
    // one-position pipeline SLOWER while the state matrix held a byte per state: the kernel was then bound by memory traffic, and more
    // loads in flight only queued.)  Descriptors are read three positions ahead; both arrays have three spare entries.
    LwSmallWords cur = steps[0], nxt = steps[1], nn = steps[2];
    int ev_cur = ev_topo[0], ev_nxt = ev_topo[1], ev_nn = ev_topo[2];
    Bytes byt, byt1;   // the parents' bytes of `cur` and of `nxt`
    Tabs tab0;         // the table lines of `cur`
    fetch_bytes(cur, byt);
    fetch_tabs(cur, tab0);
    fetch_bytes(nxt, byt1);
    uint32_t out[S];   // the ++ outputs of the last even position's step: their top halves decide it, the bottom halves the odd position after it
#pragma unroll
    for (int r = 0; r < S; ++r) out[r] = 0;
    auto position = [&](int t, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        asm volatile("" : "+v"(c32));
        const LwSmallWords sd = cur;   // a, b: the four parents' rows; c: own row, the table's base; d: the table's bytes, (flags), CPT offset, kv | parents << 4 | a1 << 8 | a2 << 16 | a3 << 24 and the pipeline's flags
        const int ev = ev_cur;
        const Bytes mine = byt;
        const Tabs mine_tab = tab0;
        const LwSmallWords sd1 = nxt, sd2 = nn;
        const LwSmallWords sd3 = steps[t + 3];
        const int ev3 = ev_topo[t + 3];
        Tabs tab1;
        Bytes byt2;
        fetch_tabs(sd1, tab1);
        fetch_bytes(sd2, byt2);
        const int kv = int(sd.d.w & 0xfu);
        const uint64_t coff = sd.d.z;
        const uint32_t a1 = (sd.d.w >> 8) & 0xfu, a2 = (sd.d.w >> 16) & 0xfu, a3 = sd.d.w >> 24;
        // the row numbers of the thread's four samples as the four bytes of one register: mixed radix over the parents, first most significant
        const uint32_t w0 = spread(mine.b0), w1 = spread(mine.b1), w2 = spread(mine.b2), w3 = spread(mine.b3);
        uint32_t rp;
        if (POW2) rp = ((((((w0 << a1) + w1) << a2) + w2) << a3)) + w3;
        else rp = ((w0 * a1 + w1) * a2 + w2) * a3 + w3;
        my_tab[lane] = mine_tab.q0;
        my_tab[64 + lane] = mine_tab.q1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t row[S];
#pragma unroll
        for (int r = 0; r < S; ++r) row[r] = (rp >> (8 * r)) & 0xffu;
        if (PAR == 0) {   // the step of this position and the next, evidence node or not
#pragma unroll
            for (int r = 0; r < S; ++r) out[r] = xoshiro_next(rng[r]);
        }
        uint32_t packed = 0;
        if (!REJECT && ev >= 0) {   // evidence node: w *= cpt[row][ev] (:148-153)
            double x[S];
#pragma unroll
            for (int r = 0; r < S; ++r) x[r] = cpt[coff + uint64_t(row[r]) * kv + ev];
#pragma unroll
            for (int r = 0; r < S; ++r) w[r] *= x[r];
            packed = uint32_t(ev) * 0x55u;
        } else {
            uint2 e[S];
#pragma unroll
            for (int r = 0; r < S; ++r) e[r] = tab[row[r]];
            int st[S];
            // state = how many thresholds the draw exceeds; a tie = some threshold equals it.  Both from the three differences
            // e_i - h (count_and_tie): seven vector instructions per sample, where three compares for ">", two selects, an add and
            // three compares for "==" (+ their scalar ORs) were nine vector and three scalar
            uint32_t nearest = 0xffffffffu;
#pragma unroll
            for (int r = 0; r < S; ++r) {
                uint32_t c, nr;
                count_and_tie<PAR>(e[r].x, e[r].y, out[r], c, nr);
                st[r] = int(c);
                nearest = min(nearest, nr);
            }
            const bool tie = nearest == 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (the next position's copy comes after these reads)
            if (__any(tie)) {   // ~1 % of a wave's positions: the four draws again, against the full thresholds
#pragma unroll
                for (int r = 0; r < S; ++r) {
                    const unsigned long long U = unit53<PAR>(half_of<PAR>(out[r]), rng[r]);
                    const unsigned long long* rowp = thr + coff + uint64_t(row[r]) * kv;
                    int c = 0;
                    for (int i = 0; i + 1 < kv; ++i) c += (U >= rowp[i]) ? 1 : 0;
                    st[r] = c;
                }
            }
#pragma unroll
            for (int r = 0; r < S; ++r) {
                if (REJECT && ev >= 0 && st[r] != ev) w[r] = 0.0;  // rejected (rejection_sampling.hpp:70-84)
                packed |= uint32_t(st[r]) << (2 * r);
            }
        }
        asm volatile("" : "+v"(c32));
        *reinterpret_cast<lw_global_u8>(row_at(sd.c.x, sd.c.y) + c32) = uint8_t(packed);
        // a later position that reads THIS node: the byte it loaded ahead is older than the store above
        if (sd1.d.w & 0x80u) patch(byt1, (sd1.d.w >> 12) & 3u, packed);        // the next position's parent j
        if (sd2.d.w & 0x400000u) patch(byt2, (sd2.d.w >> 20) & 3u, packed);    // ... the one after it
        byt = byt1; byt1 = byt2; tab0 = tab1;
        cur = sd1; ev_cur = ev_nxt;
        nxt = sd2; ev_nxt = ev_nn;
        nn = sd3; ev_nn = ev3;
    };
    int t = 0;
    for (; t + 1 < n; t += 2) {
        position(t, std::integral_constant<int, 0>());
        position(t + 1, std::integral_constant<int, 1>());
    }
    if (t < n) position(t, std::integral_constant<int, 0>());
    // (the address from c32, whose value the loop keeps opaque: formed here, not before the loop and carried through it in scratch)
    asm volatile("" : "+v"(c32));
#pragma unroll
    for (int r = 0; r < S; ++r) weights[uint64_t(c32) * S + r] = w[r];
}

// acc[i] += w on the lanes whose state is i: the compare writes exec itself (v_cmpx), the fp64 add runs under it, exec is set back to
// all lanes (the kernel runs with every lane enabled: no divergent exit before, lanes past the last node accumulate into registers
// nobody reads).  2 vector + 1 scalar instruction per state and sample; compare into vcc + s_and_saveexec + add + s_mov was 2 + 2, and
// the kernel, 9 vector and 8 scalar instructions per 64 node-samples, ran at 69 % of its vector-issue floor.
template <int KMAX>
__device__ __forceinline__ void masked_add(double (&acc)[KMAX], int i, double w, uint32_t st) {
    asm volatile(
        "v_cmpx_eq_u32_e32 vcc, %2, %1\n\t"
        "v_add_f64 %0, %0, %3\n\t"
        "s_mov_b64 exec, -1"
        : "+v"(acc[i])
        : "v"(st), "s"(uint32_t(i)), "s"(w)
        : "vcc");
}

// hist[v][state_v] += w over samples [range*R, (range+1)*R) of the first n_valid; lane = node.
template <int KMAX>
__global__ __launch_bounds__(kLwThreads) void lw_hist_kernel(const uint8_t* __restrict__ states,
                                                             const double* __restrict__ weights,
                                                             const int32_t* __restrict__ k,
                                                             const int64_t* __restrict__ node_off,
                                                             double* __restrict__ hist, int32_t n, uint64_t batch,
                                                             uint64_t n_valid, uint64_t range) {
    const int v = blockIdx.x * kLwThreads + threadIdx.x;
    const uint64_t s0 = uint64_t(blockIdx.y) * range;
    const uint64_t s1 = s0 + range < n_valid ? s0 + range : n_valid;
    if (s0 >= s1) return;
    const bool live = v < n;
    const uint8_t* rowp = states + uint64_t(live ? v : 0) * batch;
    double acc[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) acc[i] = 0.0;
    // the weights are read-only for this kernel and wave-uniform: constant address space -> s_load
    typedef double double8 __attribute__((ext_vector_type(8)));
    typedef const double8 __attribute__((address_space(4))) * const_double8s;
    uint64_t s = s0;
    // 128 samples (one 128-byte line of the lane's row) per trip, the next line in flight while this one is accumulated (two
    // register sets, no copies).  s0 and the row stride are multiples of 128.  (64 bytes per trip until round 4: the two halves of a
    // line were then requested a trip -- ~8 us -- apart, and the L2 had dropped the line in between: TCC_MISS = TCC_REQ, 2.1 x the
    // matrix fetched from memory.)
    constexpr int NQ = 8;
    auto fetch = [&](uint4 (&q)[NQ], uint64_t at) {
#pragma unroll
        for (int c = 0; c < NQ; ++c) q[c] = *reinterpret_cast<const uint4*>(rowp + at + 16 * c);
    };
    auto consume = [&](const uint4 (&q)[NQ], uint64_t at) {
        // the 16 weights of a chunk are requested while the chunk before it is accumulated (scalar loads, two chunks' worth of SGPRs)
        double8 wlo = ((const_double8s)(weights + at))[0], whi = ((const_double8s)(weights + at))[1];
#pragma unroll
        for (int c = 0; c < NQ; ++c) {
            __builtin_amdgcn_sched_barrier(0);  // keep the scalar weight loads of a chunk with the chunk before it
            const const_double8s wp = (const_double8s)(weights + at + 16 * (c < NQ - 1 ? c + 1 : NQ - 1));
            const double8 nlo = wp[0], nhi = wp[1];
            const uint32_t word[4] = {q[c].x, q[c].y, q[c].z, q[c].w};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double wj = j < 8 ? wlo[j & 7] : whi[j & 7];
                const uint32_t st = (word[j >> 2] >> (8 * (j & 3))) & 0xffu;
#pragma unroll
                for (int i = 0; i < KMAX; ++i) masked_add<KMAX>(acc, i, wj, st);
            }
            wlo = nlo;
            whi = nhi;
        }
    };
    // the segment after the last one is fetched too (clamped to the last): no branch around the
    // prefetch, so the wait before the first use can leave those four loads outstanding
    constexpr uint64_t SEG = 16 * NQ;
    const uint64_t n_seg = (s1 - s0) / SEG;
    uint4 qa[NQ], qb[NQ];
    if (n_seg > 0) fetch(qa, s0);
    for (uint64_t g = 0; g < n_seg; g += 2) {
        fetch(qb, s0 + SEG * (g + 1 < n_seg ? g + 1 : n_seg - 1));
        consume(qa, s0 + SEG * g);
        if (g + 1 >= n_seg) break;
        fetch(qa, s0 + SEG * (g + 2 < n_seg ? g + 2 : n_seg - 1));
        consume(qb, s0 + SEG * (g + 1));
    }
    s = s0 + SEG * n_seg;
    for (; s < s1; ++s) {
        const double wj = weights[s];
        const uint32_t st = rowp[s];
#pragma unroll
        for (int i = 0; i < KMAX; ++i) masked_add<KMAX>(acc, i, wj, st);
    }
    if (!live) return;
    const int kv = k[v];
    double* h = hist + node_off[v];
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
        if (i < kv && acc[i] != 0.0) atomicAdd(h + i, acc[i]);
}

// The same pass over a state matrix of TWO bits per state (LwArgs::packed2: four samples per byte): a 128-byte line of a lane's row is
// 512 samples.  Per 16 samples one dword of states and sixteen wave-uniform weights (two scalar 64-byte loads, requested while the
// dword before is accumulated).  s0 and the ranges are multiples of 512 samples, the row stride a multiple of 128 bytes.
#ifndef BN_LW_HIST2_NQ
#define BN_LW_HIST2_NQ 4
#endif
template <int KMAX>
__global__ __launch_bounds__(kLwThreads) void lw_hist2_kernel(const uint8_t* __restrict__ states, const double* __restrict__ weights,
                                                              const int32_t* __restrict__ k, const int64_t* __restrict__ node_off,
                                                              double* __restrict__ hist, int32_t n, uint64_t stride, uint64_t n_valid,
                                                              uint64_t range) {
    const int v = blockIdx.x * kLwThreads + threadIdx.x;
    const uint64_t s0 = uint64_t(blockIdx.y) * range;
    const uint64_t s1 = s0 + range < n_valid ? s0 + range : n_valid;
    if (s0 >= s1) return;
    const bool live = v < n;
    const uint8_t* rowp = states + uint64_t(live ? v : 0) * stride;
    double acc[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) acc[i] = 0.0;
    typedef double double8 __attribute__((ext_vector_type(8)));
    typedef const double8 __attribute__((address_space(4))) * const_double8s;
    constexpr int NQ = BN_LW_HIST2_NQ;    // 16-byte pieces per trip
    constexpr uint64_t SEG = 64 * NQ;     // samples per trip
    auto fetch = [&](uint4 (&q)[NQ], uint64_t at) {   // `at`: first sample of the trip
#pragma unroll
        for (int c = 0; c < NQ; ++c) q[c] = *reinterpret_cast<const uint4*>(rowp + (at >> 2) + 16 * c);
    };
    auto consume = [&](const uint4 (&q)[NQ], uint64_t at) {
        double8 wlo = ((const_double8s)(weights + at))[0], whi = ((const_double8s)(weights + at))[1];
#pragma unroll
        for (int ch = 0; ch < 4 * NQ; ++ch) {   // one dword = 16 samples
            __builtin_amdgcn_sched_barrier(0);  // keep the scalar weight loads of a chunk with the chunk before it
            const const_double8s wp = (const_double8s)(weights + at + 16 * (ch < 4 * NQ - 1 ? ch + 1 : 4 * NQ - 1));
            const double8 nlo = wp[0], nhi = wp[1];
            const uint4 qq = q[ch >> 2];
            const uint32_t word = (ch & 3) == 0 ? qq.x : (ch & 3) == 1 ? qq.y : (ch & 3) == 2 ? qq.z : qq.w;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double wj = j < 8 ? wlo[j & 7] : whi[j & 7];
                const uint32_t st = (word >> (2 * j)) & 3u;
#pragma unroll
                for (int i = 0; i < KMAX; ++i) masked_add<KMAX>(acc, i, wj, st);
            }
            wlo = nlo;
            whi = nhi;
        }
    };
    const uint64_t n_seg = (s1 - s0) / SEG;
    uint4 qa[NQ], qb[NQ];
    if (n_seg > 0) fetch(qa, s0);
    for (uint64_t g = 0; g < n_seg; g += 2) {
        fetch(qb, s0 + SEG * (g + 1 < n_seg ? g + 1 : n_seg - 1));
        consume(qa, s0 + SEG * g);
        if (g + 1 >= n_seg) break;
        fetch(qa, s0 + SEG * (g + 2 < n_seg ? g + 2 : n_seg - 1));
        consume(qb, s0 + SEG * (g + 1));
    }
    for (uint64_t s = s0 + SEG * n_seg; s < s1; ++s) {
        const double wj = weights[s];
        const uint32_t st = (uint32_t(rowp[s >> 2]) >> (2 * (s & 3))) & 3u;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) masked_add<KMAX>(acc, i, wj, st);
    }
    if (!live) return;
    const int kv = k[v];
    double* h = hist + node_off[v];
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
        if (i < kv && acc[i] != 0.0) atomicAdd(h + i, acc[i]);
}

// Any arity: a block walks the nodes, thread = sample (4 per thread), LDS histogram per node.
__global__ __launch_bounds__(kLwThreads) void lw_hist_wide_kernel(const uint8_t* __restrict__ states,
                                                                  const double* __restrict__ weights,
                                                                  const int32_t* __restrict__ k,
                                                                  const int64_t* __restrict__ node_off,
                                                                  double* __restrict__ hist, int32_t n,
                                                                  uint64_t batch, uint64_t n_valid) {
    __shared__ double sh_hist[256];
    const int tid = threadIdx.x;
    const uint64_t col = (uint64_t(blockIdx.x) * kLwThreads + tid) * kLwPerThread;
    double w[kLwPerThread];
    bool valid[kLwPerThread];
#pragma unroll
    for (int r = 0; r < kLwPerThread; ++r) {
        valid[r] = (col + r) < n_valid;
        w[r] = valid[r] ? weights[col + r] : 0.0;
    }
    for (int v = 0; v < n; ++v) {
        const int kv = k[v];
        __syncthreads();
        if (tid < kv) sh_hist[tid] = 0.0;
        __syncthreads();
        const uint32_t word = *reinterpret_cast<const uint32_t*>(states + uint64_t(v) * batch + col);
#pragma unroll
        for (int r = 0; r < kLwPerThread; ++r)
            if (valid[r] && w[r] != 0.0) unsafeAtomicAdd(&sh_hist[(word >> (8 * r)) & 0xffu], w[r]);  // ds_add_f64
        __syncthreads();
        if (tid < kv) {
            const double x = sh_hist[tid];
            if (x != 0.0) atomicAdd(&hist[node_off[v] + tid], x);
        }
    }
}

// bn_lw_states: the [node][sample] byte matrix of the last batch, transposed to sample-major [sample][node] (what make_samples
// counts joint patterns from, likelihood_weighting.hpp:62-118) through a 64 x 64 LDS tile: coalesced reads along the samples of a
// node, coalesced writes along the nodes of a sample.  (One strided copy per NODE before: 10 000 copy commands on config 5.)
template <bool PACKED2>   // PACKED2: four samples per byte in `states`, two bits each; `out` is one byte per state either way
__global__ __launch_bounds__(256) void lw_transpose_kernel(const uint8_t* __restrict__ states, uint8_t* __restrict__ out, int32_t n,
                                                           uint64_t batch, uint64_t n_samples) {
    __shared__ uint8_t tile[64][65];
    const uint64_t s0 = uint64_t(blockIdx.x) * 64;
    const int v0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    for (int r = ty; r < 64; r += 4) {   // row r of the tile = node v0 + r, column tx = sample s0 + tx
        const int v = v0 + r;
        const uint64_t smp = s0 + tx;
        uint8_t x = 0;
        if (v < n && smp < n_samples)
            x = PACKED2 ? uint8_t((states[uint64_t(v) * batch + (smp >> 2)] >> (2 * (smp & 3))) & 3u) : states[uint64_t(v) * batch + smp];
        tile[r][tx] = x;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {   // row r of the output tile = sample s0 + r, column tx = node v0 + tx
        const uint64_t smp = s0 + r;
        const int v = v0 + tx;
        if (v < n && smp < n_samples) out[smp * uint64_t(n) + v] = tile[tx][r];
    }
}
int launch_lw_transpose(const uint8_t* states, uint8_t* out, int32_t n, uint64_t batch, bool packed2, uint64_t n_samples, void* stream) {
    (void)hipGetLastError();
    if (n_samples == 0 || n <= 0) return 0;
    const dim3 grid(unsigned((n_samples + 63) / 64), unsigned((n + 63) / 64));
    if (packed2) hipLaunchKernelGGL(lw_transpose_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, states, out, n, batch, n_samples);
    else hipLaunchKernelGGL(lw_transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, states, out, n, batch, n_samples);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

int launch_lw_sample(const LwArgs& a, int blocks, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
#define BN_LW_LAUNCH3(R24, INL, REJ)                                                                              \
    hipLaunchKernelGGL((lw_sample_kernel<R24, INL, REJ>), dim3(blocks), dim3(kLwThreads), 0, (hipStream_t)stream,  \
                       reinterpret_cast<const LwStepWords*>(a.steps), reinterpret_cast<const uint4*>(a.parents),   \
                       a.ev_topo, a.cpt, a.thr, a.thr32, reinterpret_cast<const uint4*>(a.thr16), a.states, a.weights, a.n, a.batch, a.sample_base, a.seed)
#define BN_LW_LAUNCH(R24, INL)                                \
    do {                                                      \
        if (a.mode == 1) BN_LW_LAUNCH3(R24, INL, true);       \
        else BN_LW_LAUNCH3(R24, INL, false);                  \
    } while (0)
#define BN_LW_SMALL(P2, REJ)                                                                                           \
    hipLaunchKernelGGL((lw_sample_small_kernel<P2, REJ>), dim3(blocks), dim3(kLwThreads), 0, (hipStream_t)stream,       \
                       reinterpret_cast<const LwSmallWords*>(a.small_steps), a.ev_topo, a.cpt, a.thr,                   \
                       reinterpret_cast<const uint4*>(a.thr16), a.states, a.weights, a.n, a.sample_base, a.seed)
    if (a.small) {
        if (a.small_pow2 && a.mode == 1) BN_LW_SMALL(true, true);
        else if (a.small_pow2) BN_LW_SMALL(true, false);
        else if (a.mode == 1) BN_LW_SMALL(false, true);
        else BN_LW_SMALL(false, false);
    } else if (a.rows24 && a.inline_parents) BN_LW_LAUNCH(true, true);
    else if (a.rows24) BN_LW_LAUNCH(true, false);
    else if (a.inline_parents) BN_LW_LAUNCH(false, true);
    else BN_LW_LAUNCH(false, false);
#undef BN_LW_LAUNCH3
#undef BN_LW_SMALL
#undef BN_LW_LAUNCH
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

int launch_lw_hist(const LwArgs& a, int blocks, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    hipStream_t st = (hipStream_t)stream;
    if (a.kmax > 8) {
        hipLaunchKernelGGL(lw_hist_wide_kernel, dim3(blocks), dim3(kLwThreads), 0, st, a.states, a.weights, a.k,
                           a.node_off, a.hist, a.n, a.batch, a.n_valid);
    } else if (a.n_valid > 0) {
        // enough (node block, sample range) pairs to fill the chip; ranges are multiples of 256 (a block walks its range 64 samples
        // at a time: with multiples of 1024 the reference's default of 10 000 samples on a 37-node network was 10 blocks of 16
        // trips each, 70 us -- more than the sampling itself); two-bit states: multiples of 512 (a trip is a 128-byte line)
        const unsigned xb = unsigned((a.n + kLwThreads - 1) / kLwThreads);
        const uint64_t gran = a.packed2 ? 512 : 256;
        uint64_t ranges = (4096 + xb - 1) / xb;
        uint64_t range = (a.n_valid + ranges - 1) / ranges;
        range = std::max<uint64_t>(range, (a.n_valid + 1023) / 1024);  // ... and at most ~1 000 ranges: every range ends in atomics on the same bins
        range = (range + gran - 1) / gran * gran;
        const unsigned yb = unsigned((a.n_valid + range - 1) / range);
        const dim3 grid(xb, yb);
        if (a.packed2 && a.kmax <= 2)
            hipLaunchKernelGGL(lw_hist2_kernel<2>, grid, dim3(kLwThreads), 0, st, a.states, a.weights, a.k, a.node_off,
                               a.hist, a.n, a.batch, a.n_valid, range);
        else if (a.packed2)
            hipLaunchKernelGGL(lw_hist2_kernel<4>, grid, dim3(kLwThreads), 0, st, a.states, a.weights, a.k, a.node_off,
                               a.hist, a.n, a.batch, a.n_valid, range);
        else if (a.kmax <= 2)
            hipLaunchKernelGGL(lw_hist_kernel<2>, grid, dim3(kLwThreads), 0, st, a.states, a.weights, a.k, a.node_off,
                               a.hist, a.n, a.batch, a.n_valid, range);
        else if (a.kmax <= 4)
            hipLaunchKernelGGL(lw_hist_kernel<4>, grid, dim3(kLwThreads), 0, st, a.states, a.weights, a.k, a.node_off,
                               a.hist, a.n, a.batch, a.n_valid, range);
        else
            hipLaunchKernelGGL(lw_hist_kernel<8>, grid, dim3(kLwThreads), 0, st, a.states, a.weights, a.k, a.node_off,
                               a.hist, a.n, a.batch, a.n_valid, range);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

}  // namespace bnmi

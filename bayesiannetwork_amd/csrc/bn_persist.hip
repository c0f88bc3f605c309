// bn_persist.hip -- persistent dataflow variant of the BP sweep for networks that fit the chip.
//
// The per-sweep launch (bn_kernels.hip) streams every CPT from memory each iteration and pays a
// kernel boundary per iteration.  MI355X has 256 CUs x 512 KiB of vector registers: when every
// tile (wavefront) of the network can be resident at once, ONE launch runs the whole
// belief-propagation loop (belief_propagation.hpp:75-148) with
//   * the tile's CPT image loaded ONCE and kept in VGPRs for every iteration,
//   * pi(v) / lambda(v) kept in registers between iterations (written out only for the beliefs),
//   * no grid barrier: the schedule is Jacobi, so tile T may run iteration s+1 as soon as the
//     tiles that hold parents / children of its nodes finished iteration s (both the RAW and the
//     WAR hazard of the double-buffered message records); each tile publishes a counter,
//   * messages exchanged through write-through (sc1) stores and sc1 loads + relaxed agent-scope
//     flags -- the placement-independent hand-off of cdna_hip_programming.md Guideline 16 (R1);
//     no assumption on dispatch order or XCD placement,
//   * convergence with one iteration of slack: before starting iteration t a tile only needs
//     iterations <= t-2 settled (the last tile to finish an iteration reduces that iteration's
//     per-tile residuals and publishes `conv`), so nobody waits for the global decision; a tile
//     may run at most ONE iteration past convergence, which only writes the non-current buffers.
// Arithmetic and operation order are those of tile_uniform (bit-identical results, asserted by
// the parity tests, which run both paths).  Every wait is bounded by a wall-clock timeout that
// raises `abort`; the host then falls back to the per-sweep launch path.
//
// STATUS (measured on MI355X): correct -- bit-identical to the launch path over
// repeated full-size runs -- but SLOWER, so it is opt-in (bn_set_option "persistent" / BN_PERSISTENT=1):
// an iteration is a chain of dependent memory-side round trips (poll neighbours -> sc1 loads ->
// arithmetic -> drain stores -> publish), 15.6 us per iteration on a 66-tile grid against 8 us for
// a whole per-sweep launch, and >100 us at 1562 tiles where the polling itself contends with the
// traffic (cdna_hip_programming.md 5.6: hand-offs cost microseconds per hop; cut at the seam).
#include <hip/hip_runtime.h>

#include <cfloat>

#include "bn_device.hpp"

namespace bnmi {

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__host__ __device__ constexpr int ipow_p(int b, int e) { return e == 0 ? 1 : b * ipow_p(b, e - 1); }
__device__ __forceinline__ double res_acc_p(double md, double d) { return (md < d) ? d : md; }  // std::max, NaN dropped

template <int K>
__device__ __forceinline__ void normalize_p(double (&t)[K]) {  // :298-311, no zero guard
    double sum = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) sum += t[i];
#pragma unroll
    for (int i = 0; i < K; ++i) t[i] /= sum;
}

__device__ __forceinline__ unsigned long long wave_umax_p(unsigned long long x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(x, off, 64);
        x = o > x ? o : x;
    }
    return x;
}

// 16-byte write-through store / L1-bypassing load through a buffer descriptor (aux 16 = sc1);
// tracked by the compiler's s_waitcnt bookkeeping like any other load.
__device__ __forceinline__ double2_t ld_sc1(__amdgpu_buffer_rsrc_t r, int64_t idx2) {
    return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(r, int(idx2 * 16), 0, 16));
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, int64_t idx2, double2_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, int(idx2 * 16), 0, 16);
}

struct ChildRef {  // one out-edge, double2 indices (bn_plan.hpp MsgRef decoded once)
    int pi, lam, stride;
    bool has;
};

// Bounded waits.  wall_clock64() is the 100 MHz constant clock.
__device__ __forceinline__ bool timed_out(const PersistArgs& a, unsigned long long t0) {
    return wall_clock64() - t0 > a.timeout_ticks;
}

template <int K, int M, int RC>
__device__ __forceinline__ void tile_persist(const PersistArgs& a, const TileDesc& td, int tile, int lane) {
    constexpr int KP = (K + 1) & ~1, H = KP / 2;
    constexpr int C = ipow_p(K, M), S = K * C, SP = (S + 1) & ~1;
    constexpr int CB = (M > 0) ? C / K : 0;
    const BpBuffers& b = a.b;
    const bool active = lane < td.n_nodes;
    const int lc = active ? lane : 0;  // idle lanes shadow lane 0 and store nothing
    const unsigned long long t0 = wall_clock64();

    // ---- resident state: CPT, evidence mark, references, pi(v), lambda(v)
    const double2_t* cp = reinterpret_cast<const double2_t*>(b.cpt + td.cpt_base) + lc;
    double cpt[SP];
#pragma unroll
    for (int q = 0; q < SP / 2; ++q) {
        const double2_t x = cp[q * kWave];
        cpt[2 * q] = x.x;
        cpt[2 * q + 1] = x.y;
    }
    const bool frozen = b.frozen[td.slot_base + lc] != 0;
    const int64_t rbase = td.rec_base / 2 + lc;
    const MsgRef* orf = b.out_refs + td.out_base + lc;
    ChildRef oref[RC > 0 ? RC : 1];
#pragma unroll
    for (int c = 0; c < RC; ++c) {
        MsgRef r{-1, 0};
        if (c < td.cmax) r = orf[c * kWave];
        oref[c].has = r.pi >= 0;
        oref[c].pi = oref[c].has ? r.pi : 0;
        oref[c].lam = oref[c].has ? r.lam : 0;
        oref[c].stride = oref[c].has ? (r.lam - r.pi) / H : 0;  // unsharded: every record is tile-resident
    }
    double piv[KP], lav[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) {  // initial state (:38-64): roots start from their CPT row
        piv[i] = (M == 0 && i < K) ? cpt[i] : 1.0;
        lav[i] = 1.0;
    }
    if (frozen) {  // evidence nodes hold their vector as pi and lambda (:68-73)
        const double2_t* nin = reinterpret_cast<const double2_t*>(b.node0 + td.node_base) + lc;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const double2_t x = nin[h * kWave], y = nin[(H + h) * kWave];
            piv[2 * h] = x.x; piv[2 * h + 1] = x.y;
            lav[2 * h] = y.x; lav[2 * h + 1] = y.y;
        }
    }
    const size_t rec_bytes = size_t(b.rec_total_doubles) * 8;
    const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(b.rec0, 0, int(rec_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(b.rec1, 0, int(rec_bytes), 0x00020000);
    const int nbr0 = a.nbr_ptr[tile], n_nbr = a.nbr_ptr[tile + 1] - nbr0;
    const int my_nbr = lane < n_nbr ? a.nbr_idx[nbr0 + lane] : tile;  // first 64 neighbours: one per lane

    for (int s = 0;; ++s) {
        // ---- (A) one iteration of slack on the global decision: iterations <= s-2 are settled
        if (s >= 2) {
            while (__hip_atomic_load(&a.sync->completed, RLX_AGENT) < unsigned(s - 1)) {
                if (__hip_atomic_load(&a.sync->abort, RLX_AGENT) != 0) return;
                if (timed_out(a, t0)) { __hip_atomic_store(&a.sync->abort, 1u, RLX_AGENT); return; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        const unsigned conv = __hip_atomic_load(&a.sync->conv, RLX_AGENT);
        if (conv != 0 && unsigned(s) >= conv) return;  // finished (converged or max_sweeps)
        // ---- (B) data dependences: the neighbouring tiles finished iteration s-1
        if (s >= 1) {
            for (;;) {
                bool ok = __hip_atomic_load(&a.flags[my_nbr], RLX_AGENT) >= unsigned(s);
                for (int q = lane + kWave; q < n_nbr; q += kWave)  // tiles with more than 64 neighbours
                    ok = ok && __hip_atomic_load(&a.flags[a.nbr_idx[nbr0 + q]], RLX_AGENT) >= unsigned(s);
                if (__all(ok)) break;
                if (__hip_atomic_load(&a.sync->abort, RLX_AGENT) != 0) return;
                if (timed_out(a, t0)) { __hip_atomic_store(&a.sync->abort, 2u, RLX_AGENT); return; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        const bool first = s == 0;
        const __amdgpu_buffer_rsrc_t rin = (s & 1) ? rsrc1 : rsrc0;
        const __amdgpu_buffer_rsrc_t rout = (s & 1) ? rsrc0 : rsrc1;
        double* node_out = ((s & 1) ? b.node0 : b.node1) + td.node_base;

        // ---- (C) inputs: parents' pi-messages, children's lambda-messages (sc1: written by other CUs)
        double pim[M > 0 ? M : 1][KP];
        double lkc[RC > 0 ? RC : 1][KP];
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i < KP; ++i) pim[j][i] = 1.0;
#pragma unroll
        for (int c = 0; c < RC; ++c)
#pragma unroll
            for (int i = 0; i < KP; ++i) lkc[c][i] = 1.0;
        if (!first) {
#pragma unroll
            for (int j = 0; j < M; ++j)
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const double2_t x = ld_sc1(rin, rbase + ((j * 2 + 0) * H + h) * kWave);
                    pim[j][2 * h] = x.x; pim[j][2 * h + 1] = x.y;
                }
#pragma unroll
            for (int c = 0; c < RC; ++c)
                if (c < td.cmax) {
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const double2_t y = ld_sc1(rin, oref[c].lam + h * oref[c].stride);
                        lkc[c][2 * h] = oref[c].has ? y.x : 1.0;
                        lkc[c][2 * h + 1] = oref[c].has ? y.y : 1.0;
                    }
                }
        }

        // ---- (D) child role, calculate_pi (:174-200) and calculate_lambda_k (:240-266); the same
        // accumulation order as tile_uniform (bn_kernels.hip)
        double pin[K];
        double out[M > 0 ? M : 1][K];
#pragma unroll
        for (int jt = 0; jt < M; ++jt)
#pragma unroll
            for (int ct = 0; ct < K; ++ct) out[jt][ct] = 0.0;
#pragma unroll
        for (int ib = 0; ib < K; ++ib) {
            if constexpr (M == 0) {
                pin[ib] = 0.0 + cpt[ib];
            } else {
                double acc = 0.0;
                double tc[C];
#pragma unroll
                for (int c = 0; c < C; ++c) tc[c] = lav[ib] * cpt[ib * C + c];
#pragma unroll
                for (int rr = 0; rr < CB; ++rr) {
#pragma unroll
                    for (int x = 0; x < K; ++x) {
                        const int cond = rr * K + x;
                        double value = cpt[ib * C + cond];
#pragma unroll
                        for (int j = 0; j < M; ++j) value *= pim[j][(cond / ipow_p(K, M - 1 - j)) % K];
                        acc += value;
                    }
#pragma unroll
                    for (int jt = 0; jt < M; ++jt) {
                        const int stride = ipow_p(K, M - 1 - jt);
#pragma unroll
                        for (int ct = 0; ct < K; ++ct) {
                            const int cond = (rr / stride) * stride * K + ct * stride + (rr % stride);
                            double value = tc[cond];
#pragma unroll
                            for (int j = 0; j < M; ++j)
                                if (j != jt) value *= pim[j][(cond / ipow_p(K, M - 1 - j)) % K];
                            out[jt][ct] += value;
                        }
                    }
                }
                pin[ib] = acc;
            }
        }
        // parent role: lambda(v) (:220-238) and the pi-messages to the children (:202-218)
        double lan[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            double acc = 1.0;
#pragma unroll
            for (int c = 0; c < RC; ++c) acc *= lkc[c][i];
            lan[i] = acc;
        }
        normalize_p<K>(pin);
        normalize_p<K>(lan);
#pragma unroll
        for (int jt = 0; jt < M; ++jt) normalize_p<K>(out[jt]);

        // ---- (E) outputs + residual (:105-131); previous own messages re-read from the old buffer
        double wres = 0.0;
#pragma unroll
        for (int jt = 0; jt < M; ++jt) {
            double old[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i) old[i] = 1.0;
            if (!first) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const double2_t y = ld_sc1(rin, rbase + ((jt * 2 + 1) * H + h) * kWave);
                    old[2 * h] = y.x; old[2 * h + 1] = y.y;
                }
            }
            double o[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i) o[i] = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                o[i] = out[jt][i];
                wres = res_acc_p(wres, fabs(out[jt][i] - old[i]));
            }
            if (active) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    double2_t y;
                    y.x = o[2 * h]; y.y = o[2 * h + 1];
                    st_sc1(rout, rbase + ((jt * 2 + 1) * H + h) * kWave, y);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            if (c < td.cmax) {
                double u[K];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    double acc = piv[i];
#pragma unroll
                    for (int x = 0; x < RC; ++x)
                        if (x != c) acc *= lkc[x][i];
                    u[i] = acc;
                }
                normalize_p<K>(u);
                if (active && oref[c].has) {
                    double old[KP], o[KP];
#pragma unroll
                    for (int i = 0; i < KP; ++i) { old[i] = 1.0; o[i] = 0.0; }
                    if (!first) {
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            const double2_t x = ld_sc1(rin, oref[c].pi + h * oref[c].stride);
                            old[2 * h] = x.x; old[2 * h + 1] = x.y;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < K; ++i) {
                        o[i] = u[i];
                        wres = res_acc_p(wres, fabs(u[i] - old[i]));
                    }
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        double2_t y;
                        y.x = o[2 * h]; y.y = o[2 * h + 1];
                        st_sc1(rout, oref[c].pi + h * oref[c].stride, y);
                    }
                }
            }
        }
        // node vectors: registers for the next iteration, memory for the beliefs (only the finish
        // kernel reads them, after this launch); evidence nodes keep theirs (:177, :223)
        if (!frozen) {
#pragma unroll
            for (int i = 0; i < K; ++i) { piv[i] = pin[i]; lav[i] = lan[i]; }
        }
        if (active) {
            double2_t* nout = reinterpret_cast<double2_t*>(node_out) + lane;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                double2_t y, z;
                y.x = piv[2 * h]; y.y = (2 * h + 1 < K) ? piv[2 * h + 1] : 0.0;
                z.x = lav[2 * h]; z.y = (2 * h + 1 < K) ? lav[2 * h + 1] : 0.0;
                nout[h * kWave] = y;
                nout[(H + h) * kWave] = z;
            }
        }
        // ---- (F) publish: residual, then -- after every store of this wave has left -- the counter
        unsigned long long bits = active ? (unsigned long long)__double_as_longlong(wres) : 0ull;
        bits = wave_umax_p(bits);
        // one slot per tile and iteration (mod 4), written with a (write-through) STORE: a returnless
        // atomic here was observed to land after the counter below and to be missed by the reduction
        if (lane == 0) __hip_atomic_store(&a.res_tile[size_t(s & 3) * a.n_tiles + tile], bits, RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned arrived = 0;
        if (lane == 0) {
            __hip_atomic_store(&a.flags[tile], unsigned(s + 1), RLX_AGENT);
            arrived = __hip_atomic_fetch_add(&a.sync->count[s & 3], 1u, RLX_AGENT);
        }
        arrived = __shfl(arrived, 0, kWave);
        // ---- (G) the last tile to finish iteration s settles it for everybody
        if (arrived == unsigned(a.n_tiles - 1)) {
            unsigned long long m = 0;
            for (int q = lane; q < a.n_tiles; q += kWave) {
                const unsigned long long x = __hip_atomic_load(&a.res_tile[size_t(s & 3) * a.n_tiles + q], RLX_AGENT);
                m = x > m ? x : m;
            }
            m = wave_umax_p(m);
            double r = __longlong_as_double((long long)m);
            r = r < DBL_MIN ? DBL_MIN : r;  // maximum_difference starts at numeric_limits<double>::min() (:105)
            if (lane == 0) {
                if (s < b.res_cap) b.res_hist[s] = r;
                __hip_atomic_store(&a.sync->count[s & 3], 0u, RLX_AGENT);
                const bool conv_now = r < a.eps;                                   // strict '<' (:147)
                const bool capped = a.max_sweeps > 0 && s + 1 >= a.max_sweeps;
                if (conv_now || capped) {
                    b.ctl->last_res = r;
                    b.ctl->n_sweeps = s + 1;
                    b.ctl->done = conv_now ? 1 : 2;
                    __hip_atomic_store(&a.sync->conv, unsigned(s + 1), RLX_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // conv before completed
                __hip_atomic_store(&a.sync->completed, unsigned(s + 1), RLX_AGENT);
            }
        }
    }
}

template <int K, int M>
__device__ __forceinline__ void tile_persist_dispatch(const PersistArgs& a, const TileDesc& td, int tile, int lane) {
    if (td.cmax <= 2) tile_persist<K, M, 2>(a, td, tile, lane);
    else tile_persist<K, M, 4>(a, td, tile, lane);
}

__global__ __launch_bounds__(kWave, 2) void bp_persistent_kernel(PersistArgs a) {
    const int lane = threadIdx.x;
    // XCD-contiguous tile order (speed only), gridDim.x % 8 == 0
    const int nb = gridDim.x, bi = blockIdx.x;
    const int tile = (bi & 7) * (nb >> 3) + (bi >> 3);
    if (tile >= a.n_tiles) return;
    const TileDesc td = a.b.tiles[tile];
    switch (td.kv * 8 + td.m) {
        case 2 * 8 + 0: tile_persist_dispatch<2, 0>(a, td, tile, lane); break;
        case 2 * 8 + 1: tile_persist_dispatch<2, 1>(a, td, tile, lane); break;
        case 2 * 8 + 2: tile_persist_dispatch<2, 2>(a, td, tile, lane); break;
        case 2 * 8 + 3: tile_persist_dispatch<2, 3>(a, td, tile, lane); break;
        case 2 * 8 + 4: tile_persist_dispatch<2, 4>(a, td, tile, lane); break;
        case 3 * 8 + 0: tile_persist_dispatch<3, 0>(a, td, tile, lane); break;
        case 3 * 8 + 1: tile_persist_dispatch<3, 1>(a, td, tile, lane); break;
        case 3 * 8 + 2: tile_persist_dispatch<3, 2>(a, td, tile, lane); break;
        case 4 * 8 + 0: tile_persist_dispatch<4, 0>(a, td, tile, lane); break;
        case 4 * 8 + 1: tile_persist_dispatch<4, 1>(a, td, tile, lane); break;
        case 4 * 8 + 2: tile_persist_dispatch<4, 2>(a, td, tile, lane); break;
        default: __hip_atomic_store(&a.sync->abort, 3u, RLX_AGENT); break;  // not eligible: host bug
    }
}

int launch_bp_persistent(const PersistArgs& a, int grid_blocks, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    hipLaunchKernelGGL(bp_persistent_kernel, dim3(grid_blocks), dim3(kWave), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

}  // namespace bnmi

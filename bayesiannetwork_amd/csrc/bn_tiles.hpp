// bn_tiles.hpp -- device code shared by the sweep kernels: one call = one tile (one wavefront of nodes
// of one shape class, bn_plan.hpp) doing one iteration of the reference's while(true) loop
// (bayesian/inference/belief_propagation.hpp:75-148) for its nodes.  Included by the per-sweep
// kernels (bn_sweep.hpp, bn_kernels.hip) and by bn_resident.hip (the whole run in one launch).
//
// A lane (or lane group) owns everything that is computed from its node's CPT and node vectors:
//   child role  : pi(v)      = calculate_pi       (:174-200)
//                 lambda-messages v -> each parent = calculate_lambda_k (:240-266)
//   parent role : lambda(v)  = calculate_lambda   (:220-238)
//                 pi-messages v -> each child     = calculate_pi_i     (:202-218)
// All inputs come from the OLD buffers, all outputs go to the NEW buffers, exactly as the
// reference reads pi_/lambda_/pi_i_/lambda_k_ and writes new_*.  Arithmetic is fp64 in the
// reference's operation order; the library is compiled with -ffp-contract=off so the results are
// bit-identical to the C restatement in oracle/bp_oracle.c.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>

#include "bn_device.hpp"

namespace bnmi {

// Diagnostic builds only (make EXTRA=-DBN_TILE_CLOCK, scripts/experiments/tile_clock.py): lane 0 of every tile
// records the 100 MHz clock at a few points of the tile code; bn_debug_tile_clock() (bn_sweep_all.hip) reads them back.
#ifdef BN_TILE_CLOCK
constexpr int kTileClockStamps = 12, kTileClockTiles = 4096;
static __device__ unsigned long long g_tile_clock[kTileClockTiles][kTileClockStamps];
__device__ __forceinline__ void tile_stamp(int tile, int k, int lane) {
    if (lane == 0 && tile < kTileClockTiles) g_tile_clock[tile][k] = wall_clock64();
}
#define TILE_STAMP(k) tile_stamp(int(td.slot_base), (k), wlane_for_stamp)
#else
#define TILE_STAMP(k) ((void)0)
#endif


typedef double double2_t __attribute__((ext_vector_type(2)));

// NT = non-temporal output stores.  Measured on MI355X: on a working set that fits the 256 MiB
// Infinity Cache plain stores are faster (the next sweep re-reads them from cache); on an
// HBM-resident working set non-temporal stores are ~5 % faster.  The host picks per engine.
template <bool NT>
__device__ __forceinline__ void bn_store(double2_t* p, double2_t v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// The CPT image is read exactly once per sweep and never re-used inside a launch.  Measured on
// MI355X: when the working set is HBM-resident (NT policy) streaming it non-temporally is +6 %
// (it no longer evicts the message records, which two waves share, from the XCD L2); when the
// working set fits the Infinity Cache the plain load is 8 % faster.
template <bool NT>
__device__ __forceinline__ double2_t cpt_load(const double2_t* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// The four state pointers of one iteration, resolved on the host (a dynamically indexed kernarg
// array would push the whole argument struct into scratch memory).
struct IO {
    const double* rec_in;
    double* rec_out;
    const double* node_in;
    double* node_out;
    bool first;  // iteration 0: see the file comment
};

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_umax(unsigned long long x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(x, off, 64);
        x = o > x ? o : x;
    }
    return x;
}

// std::max(md, d) of libstdc++: (md < d) ? d : md -- a NaN d is dropped (:110-128)
__device__ __forceinline__ double res_acc(double md, double d) { return (md < d) ? d : md; }

template <int K>
__device__ __forceinline__ void normalize_k(double (&t)[K]) {  // :298-311, no zero guard
    double sum = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) sum += t[i];
#pragma unroll
    for (int i = 0; i < K; ++i) t[i] /= sum;
}

__host__ __device__ constexpr int ipow(int b, int e) { return e == 0 ? 1 : b * ipow(b, e - 1); }

// The residual slots of rank q inside a record buffer's exchange region (bit patterns of
// non-negative doubles; a row is accumulated with atomic umax by the sweep that WRITES the buffer).
__device__ __forceinline__ unsigned long long* res_row(const BpBuffers& b, const double* rec, int q) {
    return reinterpret_cast<unsigned long long*>(const_cast<double*>(rec)) +
           2 * (b.g_base + int64_t(q) * b.seg_d2 + b.seg_data_d2);
}

// maximum_difference of the sweep that wrote `rec`: max over every rank's slots (after the
// all-gather each rank holds all rows, so all ranks compute the same value).
__device__ __forceinline__ double reduce_residual(const BpBuffers& b, const double* rec, int lane) {
    unsigned long long m = 0;
    for (int q = 0; q < b.nranks; ++q) {
        const unsigned long long* row = res_row(b, rec, q);
#pragma unroll
        for (int i = 0; i < kResSlots / kWave; ++i) {
            unsigned long long x = __hip_atomic_load(row + i * kWave + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m = x > m ? x : m;
        }
    }
    m = wave_umax(m);
    double r = __longlong_as_double((long long)m);
    return r < DBL_MIN ? DBL_MIN : r;  // maximum_difference starts at numeric_limits<double>::min() (:105)
}

__device__ __forceinline__ void publish_residual(const BpBuffers& b, double* rec_out, int slot, double wres, int lane) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(wres);
    bits = wave_umax(bits);
    if (lane == 0 && bits != 0)
        __hip_atomic_fetch_max(res_row(b, rec_out, b.rank) + (slot & (kResSlots - 1)), bits, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

// Location of one edge's two messages (bn_plan.hpp MsgRef), double2 units from the buffer start.
struct Loc {
    int64_t pi, lam;
    int32_t stride;
    bool has;
};
__device__ __forceinline__ Loc decode_ref(MsgRef r, int h) {
    Loc l;
    l.has = r.pi >= 0;
    const bool cut = r.lam < 0;
    const int64_t lam = cut ? int64_t(~r.lam) : int64_t(r.lam);
    l.pi = l.has ? r.pi : 0;
    l.lam = l.has ? lam : 0;
    l.stride = !l.has ? 0 : (cut ? 1 : int32_t((lam - r.pi) / h));
    return l;
}

// striped element address helpers (doubles): element i of a vector whose chunks are `stride2`
// double2 apart, for node-lane nl
__device__ __forceinline__ int64_t vidx(int chunk0, int i, int stride2, int nl) {
    return int64_t(chunk0 + (i >> 1)) * (stride2 * 2) + nl * 2 + (i & 1);
}
// element i of the pi-message / lambda-message of an edge (doubles from the buffer start)
__device__ __forceinline__ int64_t pidx(const Loc& l, int i) { return (l.pi + int64_t(i >> 1) * l.stride) * 2 + (i & 1); }
__device__ __forceinline__ int64_t lidx(const Loc& l, int i) { return (l.lam + int64_t(i >> 1) * l.stride) * 2 + (i & 1); }

// ---------------------------------------------------------------------------------------------
// parent role for any shape: lambda(v) and the pi-messages to the children, operands re-read
// through L1.  Used by the generic path and by the register path when a tile has more
// children per node than it keeps in registers.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double parent_role_generic(const BpBuffers& b, const IO& io, const TileDesc& td, int kv, int kvp,
                                                     int nl, bool frozen, bool flat_cpt = false) {
    const int npt = td.npt, half = kvp >> 1;
    const double* node_in = io.node_in + td.node_base;
    double* node_out = io.node_out + td.node_base;
    const MsgRef* orf = b.out_refs + td.out_base + nl;
    const bool synth = io.first && !frozen;  // initial state instead of memory
    auto REF = [&](int c) { return decode_ref(orf[int64_t(c) * npt], half); };
    auto LK = [&](const Loc& l, int i) { return io.first ? 1.0 : io.rec_in[lidx(l, i)]; };
    auto PIV = [&](int i) {
        if (!synth) return node_in[vidx(0, i, npt, nl)];
        if (td.m != 0) return 1.0;
        // a root starts from its CPT row (:58-64); flat tiles keep entry i in lane i, slot 0
        return flat_cpt ? b.cpt[td.cpt_base + (int64_t(nl) * (kWave / npt) + i) * 2] : b.cpt[td.cpt_base + int64_t(i >> 1) * 128 + nl * 2 + (i & 1)];
    };
    double wres = 0.0;
    // lambda(v): product of the children's lambda-messages from 1.0, ascending child order (:229-235)
    if (frozen) {
        for (int i = 0; i < kv; ++i) node_out[vidx(half, i, npt, nl)] = node_in[vidx(half, i, npt, nl)];
    } else {
        double sum = 0;
        for (int i = 0; i < kv; ++i) {
            double acc = 1.0;
            for (int c = 0; c < td.cmax; ++c) {
                const Loc l = REF(c);
                if (l.has) acc *= LK(l, i);
            }
            node_out[vidx(half, i, npt, nl)] = acc;
            sum += acc;
        }
        for (int i = 0; i < kv; ++i) node_out[vidx(half, i, npt, nl)] = node_out[vidx(half, i, npt, nl)] / sum;
    }
    // pi-message to child c: pi(v) times the OTHER children's lambda-messages (:207-214)
    for (int c = 0; c < td.cmax; ++c) {
        const Loc lc = REF(c);
        if (!lc.has) continue;
        double sum = 0;
        for (int i = 0; i < kv; ++i) {
            double acc = PIV(i);
            for (int x = 0; x < td.cmax; ++x) {
                if (x == c) continue;
                const Loc lx = REF(x);
                if (lx.has) acc *= LK(lx, i);
            }
            io.rec_out[pidx(lc, i)] = acc;
            sum += acc;
        }
        for (int i = 0; i < kv; ++i) {
            const double nv = io.rec_out[pidx(lc, i)] / sum;
            io.rec_out[pidx(lc, i)] = nv;
            wres = res_acc(wres, fabs(nv - (io.first ? 1.0 : io.rec_in[pidx(lc, i)])));
        }
    }
    return wres;
}

// ---------------------------------------------------------------------------------------------
// parent role with up to RC children held in registers (k = K for node and messages): each
// child's lambda-message is loaded ONCE; the O(c^2) products of calculate_pi_i (:207-214) then run
// on registers in the reference's order (ascending child, skipping the target).  Caller passes
// pi(v) / lambda(v) as it read them (or synthesised them in iteration 0).
// ---------------------------------------------------------------------------------------------
template <int K, int RC, bool NT>
__device__ __forceinline__ double parent_role_regs(const BpBuffers& b, const IO& io, const TileDesc& td, int nl,
                                                   bool frozen, const double* piv, const double* lav) {
    constexpr int KP = (K + 1) & ~1, H = KP / 2;
    const int npt = td.npt;
    const MsgRef* orf = b.out_refs + td.out_base + nl;
    const double2_t* rec_in2 = reinterpret_cast<const double2_t*>(io.rec_in);
    double2_t* rec_out2 = reinterpret_cast<double2_t*>(io.rec_out);
    double2_t* nout = reinterpret_cast<double2_t*>(io.node_out + td.node_base) + nl;
    double wres = 0.0;
    MsgRef oref[RC];  // kept packed (8 B per child) and decoded at each use: registers matter here
    double lkc[RC][KP];
#pragma unroll
    for (int c = 0; c < RC; ++c) {
        oref[c] = MsgRef{-1, 0};
        if (c < td.cmax) oref[c] = orf[int64_t(c) * npt];
    }
#pragma unroll
    for (int c = 0; c < RC; ++c) {
#pragma unroll
        for (int i = 0; i < KP; ++i) lkc[c][i] = 1.0;
        if (c < td.cmax && !io.first) {
            const Loc l = decode_ref(oref[c], H);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const double2_t y = rec_in2[l.lam + h * l.stride];
                lkc[c][2 * h] = l.has ? y.x : 1.0;
                lkc[c][2 * h + 1] = l.has ? y.y : 1.0;
            }
        }
    }
    {   // lambda(v) (:220-238)
        double t[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            double acc = 1.0;
#pragma unroll
            for (int c = 0; c < RC; ++c) acc *= lkc[c][i];
            t[i] = acc;
        }
        normalize_k<K>(t);
        double l[KP];
#pragma unroll
        for (int i = 0; i < KP; ++i) l[i] = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i) l[i] = frozen ? lav[i] : t[i];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            double2_t z;
            z.x = l[2 * h]; z.y = l[2 * h + 1];
            bn_store<NT>(&nout[(H + h) * npt], z);
        }
    }
    // pi-message to child c (:202-218): pi(v) * the OTHER children's lambda-messages, ascending.
    // The factors before c are a running prefix shared by all later children (same multiplication
    // sequence as starting over from pi(v)); children past cmax hold 1.0 and are skipped (x * 1.0 == x).
    double pre[K];
#pragma unroll
    for (int i = 0; i < K; ++i) pre[i] = piv[i];
#pragma unroll
    for (int c = 0; c < RC; ++c) {
        if (c < td.cmax) {  // wave-uniform
            double u[K];
#pragma unroll
            for (int i = 0; i < K; ++i) u[i] = pre[i];
#pragma unroll
            for (int x = c + 1; x < RC; ++x)
                if (x < td.cmax) {
#pragma unroll
                    for (int i = 0; i < K; ++i) u[i] *= lkc[x][i];
                }
#pragma unroll
            for (int i = 0; i < K; ++i) pre[i] *= lkc[c][i];
            normalize_k<K>(u);
            const Loc l = decode_ref(oref[c], H);
            if (l.has) {
                double o[KP], old[KP];
#pragma unroll
                for (int i = 0; i < KP; ++i) { o[i] = 0.0; old[i] = 1.0; }
                if (!io.first) {
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const double2_t x = rec_in2[l.pi + h * l.stride];
                        old[2 * h] = x.x; old[2 * h + 1] = x.y;
                    }
                }
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    o[i] = u[i];
                    wres = res_acc(wres, fabs(u[i] - old[i]));
                }
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    double2_t y;
                    y.x = o[2 * h]; y.y = o[2 * h + 1];
                    bn_store<NT>(&rec_out2[l.pi + h * l.stride], y);
                }
            }
        }
    }
    return wres;
}

// parent role for a node of arity K with any number of children: registers up to 16, memory beyond
template <int K, bool NT, int RCMAX>
__device__ __forceinline__ double parent_role_any(const BpBuffers& b, const IO& io, const TileDesc& td, int nl,
                                                  bool frozen, const double* piv, const double* lav) {
    if (td.cmax <= 4) return parent_role_regs<K, 4, NT>(b, io, td, nl, frozen, piv, lav);
    if (td.cmax <= 8) return parent_role_regs<K, 8, NT>(b, io, td, nl, frozen, piv, lav);
    if constexpr (RCMAX >= 16) {
        if (td.cmax <= 16) return parent_role_regs<K, 16, NT>(b, io, td, nl, frozen, piv, lav);
    } else if constexpr (RCMAX >= 12) {
        if (td.cmax <= 12) return parent_role_regs<K, 12, NT>(b, io, td, nl, frozen, piv, lav);
    }
    return parent_role_generic(b, io, td, K, (K + 1) & ~1, nl, frozen);
}

// ---------------------------------------------------------------------------------------------
// generic tile: any arities, runtime loops, one lane per node.  Correctness path for shapes
// without a register-resident instantiation.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double tile_generic(const BpBuffers& b, const IO& io, const TileDesc& td, const ClassDesc& c, int lane) {
    double wres = 0.0;
    if (lane >= td.n_nodes) return wres;
    const double* cpt = b.cpt + td.cpt_base + lane * 2;
    const double* node_in = io.node_in + td.node_base + lane * 2;
    double* node_out = io.node_out + td.node_base + lane * 2;
    const bool frozen = b.frozen[td.slot_base + lane] == b.frozen_mark;
    const bool synth = io.first && !frozen;
    const int kv = c.kv, m = c.m, rows = c.rows, hv = c.kvp >> 1;
    // in-edge j: inside the tile's record block, or wherever the reference says (boundary tile)
    auto IN = [&](int j) {
        if (td.in_ref_base >= 0) return decode_ref(b.in_refs[td.in_ref_base + int64_t(j) * kWave + lane], c.kpp[j] >> 1);
        Loc l;
        l.has = true;
        l.pi = (td.rec_base + c.rec_off[j]) / 2 + lane;
        l.stride = kWave;
        l.lam = l.pi + int64_t(c.kpp[j] >> 1) * kWave;
        return l;
    };
    auto CPT = [&](int q) { return cpt[int64_t(q >> 1) * 128 + (q & 1)]; };
    auto PIM = [&](int j, int s) { return io.first ? 1.0 : io.rec_in[pidx(IN(j), s)]; };
    auto NIDX = [&](int part, int i) { return int64_t(part * hv + (i >> 1)) * 128 + (i & 1); };
    auto LAV = [&](int i) { return synth ? 1.0 : node_in[NIDX(1, i)]; };

    // pi(v) (:174-200): assignment ascending, value = cpt * pi-messages in ascending parent order
    if (frozen) {
        for (int i = 0; i < kv; ++i) node_out[NIDX(0, i)] = node_in[NIDX(0, i)];
    } else {
        double sum = 0;
        for (int i = 0; i < kv; ++i) {
            double acc = 0.0;
            for (int cond = 0; cond < rows; ++cond) {
                double value = CPT(i * rows + cond);
                for (int j = 0; j < m; ++j) value *= PIM(j, (cond / c.cstride[j]) % c.kp[j]);
                acc += value;
            }
            node_out[NIDX(0, i)] = acc;
            sum += acc;
        }
        for (int i = 0; i < kv; ++i) node_out[NIDX(0, i)] = node_out[NIDX(0, i)] / sum;
    }
    // lambda-message to parent jt (:240-266): for each target state, child state outer and
    // assignment inner -- the order in which the reference adds into matrix[0][cond.at(target)]
    for (int jt = 0; jt < m; ++jt) {
        const int kt = c.kp[jt];
        const Loc lt = IN(jt);
        double sum = 0;
        for (int ct = 0; ct < kt; ++ct) {
            double acc = 0.0;
            for (int i = 0; i < kv; ++i) {
                const double times = LAV(i);
                for (int cond = 0; cond < rows; ++cond) {
                    if ((cond / c.cstride[jt]) % kt != ct) continue;
                    double value = times * CPT(i * rows + cond);
                    for (int j = 0; j < m; ++j)
                        if (j != jt) value *= PIM(j, (cond / c.cstride[j]) % c.kp[j]);
                    acc += value;
                }
            }
            io.rec_out[lidx(lt, ct)] = acc;
            sum += acc;
        }
        for (int ct = 0; ct < kt; ++ct) {
            const double nv = io.rec_out[lidx(lt, ct)] / sum;
            io.rec_out[lidx(lt, ct)] = nv;
            wres = res_acc(wres, fabs(nv - (io.first ? 1.0 : io.rec_in[lidx(lt, ct)])));
        }
    }
    return res_acc(wres, parent_role_generic(b, io, td, kv, c.kvp, lane, frozen));
}

// ---------------------------------------------------------------------------------------------
// register-resident tile: node and parents share arity K, M parents, whole CPT (K^(M+1) <= 64
// doubles) in VGPRs, every loop unrolled at compile time, 16-byte lane-striped loads.
// RC = children per node held in registers (the tile's cmax <= RC; RC = 0: parent role elsewhere).
// ---------------------------------------------------------------------------------------------
// IND: boundary tile (sharded run) -- in-edge records are reached through MsgRefs; otherwise
// they are addressed arithmetically inside the tile's own block with immediate offsets.
template <int K, int M, int RC, bool NT, bool IND>
__device__ __forceinline__ double tile_uniform(const BpBuffers& b, const IO& io, const TileDesc& td, int lane) {
    constexpr int KP = (K + 1) & ~1, H = KP / 2;
    constexpr int C = ipow(K, M), S = K * C, SP = (S + 1) & ~1;
    constexpr int CB = (M > 0) ? C / K : 0;  // assignments per lambda bucket and own state
    double wres = 0.0;
    if (lane < td.n_nodes) {
#ifdef BN_TILE_CLOCK
        const int wlane_for_stamp = lane;
#endif
        TILE_STAMP(0);
        // ---- parent-role loads FIRST: the out-edge references head a dependent chain
        // (reference -> child record), so they are issued before the 32 CPT loads stream in
        const double2_t* rec_in2 = reinterpret_cast<const double2_t*>(io.rec_in);
        double2_t* rec_out2 = reinterpret_cast<double2_t*>(io.rec_out);
        // out-edge references, then the children's lambda-messages.
        // A missing child reads record 0 and contributes 1.0 (x * 1.0 == x exactly).
        const MsgRef* orf = b.out_refs + td.out_base + lane;
        Loc oref[RC > 0 ? RC : 1];
        double lkc[RC > 0 ? RC : 1][KP];
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            MsgRef r{-1, 0};
            if (c < td.cmax) r = orf[c * kWave];
            oref[c] = decode_ref(r, H);
        }

        // ---- child-role loads: CPT, pi-messages from the parents, pi(v), lambda(v)
        const double2_t* cp = reinterpret_cast<const double2_t*>(b.cpt + td.cpt_base) + lane;
        double cpt[SP];
#pragma unroll
        for (int q = 0; q < SP / 2; ++q) {
            const double2_t x = cpt_load<NT>(&cp[q * kWave]);
            cpt[2 * q] = x.x;
            cpt[2 * q + 1] = x.y;
        }
        // the children's lambda-messages (second hop of the reference chain)
#pragma unroll
        for (int c = 0; c < RC; ++c) {
#pragma unroll
            for (int i = 0; i < KP; ++i) lkc[c][i] = 1.0;
            if (c < td.cmax && !io.first) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const double2_t y = rec_in2[oref[c].lam + h * oref[c].stride];
                    lkc[c][2 * h] = oref[c].has ? y.x : 1.0;
                    lkc[c][2 * h + 1] = oref[c].has ? y.y : 1.0;
                }
            }
        }
        // in-edge j's record: inside the tile's own block (arithmetic), or -- boundary tile, some
        // parent lives on another rank -- wherever its reference says (exchange region for cut edges)
        Loc in[M > 0 ? M : 1];
        if constexpr (IND) {
#pragma unroll
            for (int j = 0; j < M; ++j) in[j] = decode_ref(b.in_refs[td.in_ref_base + j * kWave + lane], H);
        }
        const int64_t rbase = td.rec_base / 2 + lane;  // this lane's slot in the tile's record block
        // chunk h of the pi-message (part 0) / lambda-message (part 1) of in-edge j
        auto in_idx = [&](int j, int part, int h) -> int64_t {
            if constexpr (IND) return (part ? in[j].lam : in[j].pi) + h * in[j].stride;
            else return rbase + ((j * 2 + part) * H + h) * kWave;
        };
        const double2_t* nin = reinterpret_cast<const double2_t*>(io.node_in + td.node_base) + lane;
        double2_t* nout = reinterpret_cast<double2_t*>(io.node_out + td.node_base) + lane;
        const bool frozen = b.frozen[td.slot_base + lane] == b.frozen_mark;
        double pim[M > 0 ? M : 1][KP];
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i < KP; ++i) pim[j][i] = 1.0;
        if (!io.first) {
#pragma unroll
            for (int j = 0; j < M; ++j)
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const double2_t x = rec_in2[in_idx(j, 0, h)];
                    pim[j][2 * h] = x.x; pim[j][2 * h + 1] = x.y;
                }
        }
        double piv[KP], lav[KP];
#pragma unroll
        for (int i = 0; i < KP; ++i) {  // initial state (:38-64): roots start from their CPT row
            piv[i] = (M == 0 && i < K) ? cpt[i] : 1.0;
            lav[i] = 1.0;
        }
        if (!io.first || frozen) {  // evidence nodes hold their vector as pi and lambda (:68-73)
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const double2_t x = nin[h * kWave], y = nin[(H + h) * kWave];
                piv[2 * h] = x.x; piv[2 * h + 1] = x.y;
                lav[2 * h] = y.x; lav[2 * h + 1] = y.y;
            }
        }

        TILE_STAMP(1);  // loads issued
        // ---- child role.  calculate_pi (:174-200): pi[i] = sum over assignments (ascending) of
        // cpt * pi-messages (ascending parent order).  calculate_lambda_k (:240-266): bucket
        // out[jt][ct] receives, own state outer and assignment inner, (lambda[i] * cpt) * the
        // OTHER parents' pi-messages.  Each accumulator sees its terms in exactly that order;
        // the K + M*K independent chains are interleaved round-robin.
        double pin[K];
        double out[M > 0 ? M : 1][K];
#pragma unroll
        for (int jt = 0; jt < M; ++jt)
#pragma unroll
            for (int ct = 0; ct < K; ++ct) out[jt][ct] = 0.0;
#pragma unroll
        for (int ib = 0; ib < K; ++ib) {  // own state i: the CPT image is i-major
            if constexpr (M == 0) {
                pin[ib] = 0.0 + cpt[ib];
            } else {
                double acc = 0.0;
                double tc[C];  // lambda(v)[i] * cpt[cond][i], shared by the M lambda-messages
#pragma unroll
                for (int c = 0; c < C; ++c) tc[c] = lav[ib] * cpt[ib * C + c];
#pragma unroll
                for (int rr = 0; rr < CB; ++rr) {
#pragma unroll
                    for (int x = 0; x < K; ++x) {  // pi chain: assignments rr*K .. rr*K+K-1
                        const int cond = rr * K + x;
                        double value = cpt[ib * C + cond];
#pragma unroll
                        for (int j = 0; j < M; ++j) value *= pim[j][(cond / ipow(K, M - 1 - j)) % K];
                        acc += value;
                    }
#pragma unroll
                    for (int jt = 0; jt < M; ++jt) {
                        const int stride = ipow(K, M - 1 - jt);
#pragma unroll
                        for (int ct = 0; ct < K; ++ct) {  // rr-th assignment whose digit jt equals ct
                            const int cond = (rr / stride) * stride * K + ct * stride + (rr % stride);
                            double value = tc[cond];
#pragma unroll
                            for (int j = 0; j < M; ++j)
                                if (j != jt) value *= pim[j][(cond / ipow(K, M - 1 - j)) % K];
                            out[jt][ct] += value;
                        }
                    }
                }
                pin[ib] = acc;
            }
        }

        TILE_STAMP(2);  // contraction
        // ---- parent role: lambda(v) (:220-238), products in ascending child order from 1.0
        double lan[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            double acc = 1.0;
#pragma unroll
            for (int c = 0; c < RC; ++c) acc *= lkc[c][i];
            lan[i] = acc;
        }
        // ---- normalise everything (:298-311), residual (:105-131), stores
        normalize_k<K>(pin);
        normalize_k<K>(lan);
#pragma unroll
        for (int jt = 0; jt < M; ++jt) normalize_k<K>(out[jt]);
        {
            double o[KP], l[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i) { o[i] = 0.0; l[i] = 0.0; }
#pragma unroll
            for (int i = 0; i < K; ++i) {  // evidence nodes keep pi and lambda (:177, :223)
                o[i] = frozen ? piv[i] : pin[i];
                l[i] = frozen ? lav[i] : lan[i];
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                double2_t y, z;
                y.x = o[2 * h]; y.y = o[2 * h + 1];
                z.x = l[2 * h]; z.y = l[2 * h + 1];
                bn_store<NT>(&nout[h * kWave], y);
                if (RC > 0) bn_store<NT>(&nout[(H + h) * kWave], z);
            }
        }
#pragma unroll
        for (int jt = 0; jt < M; ++jt) {
            double o[KP], old[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i) { o[i] = 0.0; old[i] = 1.0; }
            if (!io.first) {
#pragma unroll
                for (int h = 0; h < H; ++h) {  // previous lambda-message of this edge, for the residual
                    const double2_t y = rec_in2[in_idx(jt, 1, h)];
                    old[2 * h] = y.x; old[2 * h + 1] = y.y;
                }
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                o[i] = out[jt][i];
                wres = res_acc(wres, fabs(out[jt][i] - old[i]));
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                double2_t y;
                y.x = o[2 * h]; y.y = o[2 * h + 1];
                bn_store<NT>(&rec_out2[in_idx(jt, 1, h)], y);
            }
        }
        TILE_STAMP(3);  // node vectors and lambda-messages normalised and stored
        // pi-message to child c (:202-218): pi(v) times the OTHER children's lambda-messages
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            if (c < td.cmax) {  // wave-uniform
                double u[K];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    double acc = piv[i];
#pragma unroll
                    for (int x = 0; x < RC; ++x)
                        if (x != c) acc *= lkc[x][i];
                    u[i] = acc;
                }
                normalize_k<K>(u);
                if (oref[c].has) {
                    double o[KP], old[KP];
#pragma unroll
                    for (int i = 0; i < KP; ++i) { o[i] = 0.0; old[i] = 1.0; }
                    if (!io.first) {
#pragma unroll
                        for (int h = 0; h < H; ++h) {  // previous pi-message of this edge, for the residual
                            const double2_t x = rec_in2[oref[c].pi + h * oref[c].stride];
                            old[2 * h] = x.x; old[2 * h + 1] = x.y;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < K; ++i) {
                        o[i] = u[i];
                        wres = res_acc(wres, fabs(u[i] - old[i]));
                    }
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        double2_t y;
                        y.x = o[2 * h]; y.y = o[2 * h + 1];
                        bn_store<NT>(&rec_out2[oref[c].pi + h * oref[c].stride], y);
                    }
                }
            }
        }
        TILE_STAMP(4);  // fused pi-messages
        if constexpr (RC == 0)  // more children than the fused path holds: separate parent role
            wres = res_acc(wres, parent_role_any<K, NT, 16>(b, io, td, lane, frozen, piv, lav));
        TILE_STAMP(5);  // separate parent role
#ifdef BN_TILE_CLOCK
        if (lane == 0 && td.slot_base < kTileClockTiles)
            g_tile_clock[td.slot_base][10] = (unsigned long long)RC | ((unsigned long long)M << 8) | ((unsigned long long)K << 16) |
                                             ((unsigned long long)td.cmax << 24) | (0xfffeull << 32);
#endif
    }
    return wres;
}

// ---------------------------------------------------------------------------------------------
// lane-group tile: k = 4, M = D + T parents, G = 4^D lanes per node (NPT = 64 / G nodes per
// wave).  Lane (nl, g) owns the assignments whose D leading parents are in state digits(g) -- the
// T trailing parents and the own state: 64 CPT entries with T = 2, 16 with T = 1 (four times the
// waves, a quarter of the serial work per wave: what a latency-bound network wants, bn_plan.cpp) --
// streams them through registers like the register-resident path, and the partial sums are combined
// inside the G-lane group with shuffles.  Products keep the reference's ascending-parent order (:190-193, :250-258); the SUM
// over assignments is re-associated across lanes, so results agree with the reference to
// rounding (not bit-for-bit; with >= 3 parents the reference's own products are unordered, :253).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double shfl_xor_d(double x, int mask) { return __shfl_xor(x, mask, kWave); }
__device__ __forceinline__ double shfl_d(double x, int src) { return __shfl(x, src, kWave); }
__device__ __forceinline__ double pick4(const double (&v)[4], int d) {
    return d == 0 ? v[0] : (d == 1 ? v[1] : (d == 2 ? v[2] : v[3]));
}

template <int D, int T, bool NT>
__device__ __forceinline__ double tile_group(const BpBuffers& b, const IO& io, const TileDesc& td, int lane) {
    constexpr int K = 4, H = 2, M = D + T;
    constexpr int G = 1 << (2 * D), NPT = kWave / G;
    constexpr int CL = 1 << (2 * T), E = K * CL;  // assignments of the T trailing parents, CPT entries per lane
    static_assert(T == 1 || T == 2, "one or two trailing parents per lane");
    const int nl = lane / G, g = lane % G;
    const bool active = nl < td.n_nodes;
    double wres = 0.0;

#ifdef BN_TILE_CLOCK
    const int wlane_for_stamp = lane;
#endif
    TILE_STAMP(0);
    // ---- loads: this lane's E CPT entries (i-major: q = i*CL + the trailing parents' states, last parent fastest)
    const double2_t* cp = reinterpret_cast<const double2_t*>(b.cpt + td.cpt_base) + lane;
    double cpt[E];
#pragma unroll
    for (int q = 0; q < E / 2; ++q) {
        const double2_t x = cpt_load<NT>(&cp[q * kWave]);
        cpt[2 * q] = x.x;
        cpt[2 * q + 1] = x.y;
    }
    const double2_t* rec_in2 = reinterpret_cast<const double2_t*>(io.rec_in);
    double2_t* rec_out2 = reinterpret_cast<double2_t*>(io.rec_out);
    const int nlc = active ? nl : 0;  // inactive groups shadow node 0 and write nothing
    Loc in[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        in[j].has = true;
        in[j].pi = td.rec_base / 2 + (j * 2 * H) * NPT + nlc;
        in[j].lam = in[j].pi + H * NPT;
        in[j].stride = NPT;
    }
    if (td.in_ref_base >= 0) {
#pragma unroll
        for (int j = 0; j < M; ++j) in[j] = decode_ref(b.in_refs[td.in_ref_base + j * NPT + nlc], H);
    }
    const double2_t* nin = reinterpret_cast<const double2_t*>(io.node_in + td.node_base) + nlc;
    double2_t* nout = reinterpret_cast<double2_t*>(io.node_out + td.node_base) + nlc;
    const bool frozen = b.frozen[td.slot_base + nlc] == b.frozen_mark;
    double pim[M][K];
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int i = 0; i < K; ++i) pim[j][i] = 1.0;
    if (!io.first) {
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const double2_t x = rec_in2[in[j].pi + h * in[j].stride];
                pim[j][2 * h] = x.x; pim[j][2 * h + 1] = x.y;
            }
    }
    double piv[K], lav[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { piv[i] = 1.0; lav[i] = 1.0; }
    if (!io.first || frozen) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const double2_t x = nin[h * NPT], y = nin[(H + h) * NPT];
            piv[2 * h] = x.x; piv[2 * h + 1] = x.y;
            lav[2 * h] = y.x; lav[2 * h + 1] = y.y;
        }
    }
    // The node is finished by M + 1 lanes of its group (G >= M + 1 for every D): lane 0 normalises pi(v),
    // lane 1 + jt the lambda-message to parent jt -- one normalisation (four divisions) per lane instead of
    // 4 (M + 1) in a row on one lane -- and each requests the previous value of ITS message (for the
    // residual, :105-131) now, with the other inputs, instead of in a round trip of its own at the end.
    static_assert(G >= M + 1, "a group has a lane per finished vector");
    Loc fin = in[0];
#pragma unroll
    for (int j = 1; j < M; ++j)
        if (g == j + 1) fin = in[j];
    const bool fin_msg = active && g >= 1 && g <= M;
    double fold[K];
#pragma unroll
    for (int i = 0; i < K; ++i) fold[i] = 1.0;
    if (fin_msg && !io.first) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const double2_t y = rec_in2[fin.lam + h * fin.stride];
            fold[2 * h] = y.x; fold[2 * h + 1] = y.y;
        }
    }
    TILE_STAMP(1);  // first-trip loads issued
    // ---- parent role (:202-238), spread over the group's lanes: lane g serves children g, g+G, ...
    // of node nl.  It needs only the OLD pi(v)/lambda(v) and the children's records, so its loads
    // ride behind the CPT stream and its arithmetic is cmax steps for the whole group instead of
    // cmax^2 on one lane.  Every product keeps the reference's ascending-child order.
    constexpr int CPL = G >= 16 ? 1 : 2;  // children per lane
    const bool par_fast = td.cmax <= G * CPL;
    if (par_fast) {
        const MsgRef* orf = b.out_refs + td.out_base + nlc;
        Loc cl[CPL];
        double clk[CPL][K], cold[CPL][K];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int c = g + q * G;
            MsgRef r{-1, 0};
            if (active && c < td.cmax) r = orf[int64_t(c) * NPT];
            cl[q] = decode_ref(r, H);
        }
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
#pragma unroll
            for (int i = 0; i < K; ++i) { clk[q][i] = 1.0; cold[q][i] = 1.0; }
            if (!io.first) {  // a missing child reads record 0 and contributes 1.0
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const double2_t y = rec_in2[cl[q].lam + h * cl[q].stride];
                    const double2_t x = rec_in2[cl[q].pi + h * cl[q].stride];
                    clk[q][2 * h] = cl[q].has ? y.x : 1.0; clk[q][2 * h + 1] = cl[q].has ? y.y : 1.0;
                    cold[q][2 * h] = x.x; cold[q][2 * h + 1] = x.y;
                }
            }
        }
        TILE_STAMP(2);  // children's records requested (the references have arrived)
        double lam_all[K], msg[CPL][K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            lam_all[i] = 1.0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) msg[q][i] = piv[i];
        }
        for (int x = 0; x < td.cmax; ++x) {  // ascending child order; wave-uniform trip count
            const int qx = x / G, src = nl * G + (x % G);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                double mine = clk[0][i];
#pragma unroll
                for (int q = 1; q < CPL; ++q) mine = (qx == q) ? clk[q][i] : mine;
                const double val = shfl_d(mine, src);
                lam_all[i] *= val;
#pragma unroll
                for (int q = 0; q < CPL; ++q)
                    if (g + q * G != x) msg[q][i] *= val;
            }
        }
        if (active && g == 0) {  // lambda(v) (:220-238)
            normalize_k<K>(lam_all);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                double2_t z;
                z.x = frozen ? lav[2 * h] : lam_all[2 * h];
                z.y = frozen ? lav[2 * h + 1] : lam_all[2 * h + 1];
                bn_store<NT>(&nout[(H + h) * NPT], z);
            }
        }
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            if (active && cl[q].has) {
                normalize_k<K>(msg[q]);
#pragma unroll
                for (int i = 0; i < K; ++i) wres = res_acc(wres, fabs(msg[q][i] - cold[q][i]));
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    double2_t y;
                    y.x = msg[q][2 * h]; y.y = msg[q][2 * h + 1];
                    bn_store<NT>(&rec_out2[cl[q].pi + h * cl[q].stride], y);
                }
            }
        }
    }

    TILE_STAMP(3);  // parent role done (the children's records have arrived)
    // pi-message entries of the lane-fixed parents
    double pfix[D];
#pragma unroll
    for (int j = 0; j < D; ++j) pfix[j] = pick4(pim[j], (g >> (2 * (D - 1 - j))) & 3);

    // ---- partial sums over this lane's E entries.  The four own states of one trailing-parent assignment are
    // handled TOGETHER: their product chains (cpt * pi-messages, ascending parent order as in the reference,
    // :190-193, :250-258) are independent of each other, so four of them are in flight where one chain alone would
    // leave a SIMD that holds no other wave waiting for its own results half the time (tile stamps, round 2: 4.2 us
    // for the 1 280 fp64 instructions of a 4-parent tile = ~50 % of one wave's issue rate).  pi(v)[i] adds its terms
    // in assignment order as before; a lambda bucket receives the sum over the own states of one assignment
    // ((t0 + t1) + (t2 + t3)) per step -- the sum over a lane's entries is re-associated, like the sum across the
    // group's lanes already is (results agree with the reference to rounding; its own >= 3-parent products are
    // unordered, :253).
    double pp[K];          // pi(v)[i]
    double ol[T][K];       // lambda-messages to the trailing parents, by target state
    double sf[D];          // lambda-messages to the leading parents: this lane's own bucket
#pragma unroll
    for (int i = 0; i < K; ++i) pp[i] = 0.0;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int ct = 0; ct < K; ++ct) ol[t][ct] = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) sf[j] = 0.0;
#pragma unroll
    for (int cl = 0; cl < CL; ++cl) {
        int ct[T];  // states of the trailing parents D .. M-1 in this assignment
#pragma unroll
        for (int t = 0; t < T; ++t) ct[t] = (cl >> (2 * (T - 1 - t))) & 3;
        double e[K], v[K], tc[K], pre[K], w[K];
#pragma unroll
        for (int ib = 0; ib < K; ++ib) e[ib] = cpt[ib * CL + cl];
        // calculate_pi: cpt * pi-messages, ascending parent order
#pragma unroll
        for (int ib = 0; ib < K; ++ib) v[ib] = e[ib];
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int ib = 0; ib < K; ++ib) v[ib] *= pfix[j];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int ib = 0; ib < K; ++ib) v[ib] *= pim[D + t][ct[t]];
#pragma unroll
        for (int ib = 0; ib < K; ++ib) pp[ib] += v[ib];
        // calculate_lambda_k: (lambda[i] * cpt) * the OTHER parents' pi-messages, ascending
#pragma unroll
        for (int ib = 0; ib < K; ++ib) tc[ib] = lav[ib] * e[ib];
#pragma unroll
        for (int ib = 0; ib < K; ++ib) pre[ib] = tc[ib];  // shared prefix over the leading parents
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int ib = 0; ib < K; ++ib) pre[ib] *= pfix[j];
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int ib = 0; ib < K; ++ib) w[ib] = pre[ib];
#pragma unroll
            for (int t2 = 0; t2 < T; ++t2)
                if (t2 != t) {
#pragma unroll
                    for (int ib = 0; ib < K; ++ib) w[ib] *= pim[D + t2][ct[t2]];
                }
            ol[t][ct[t]] += (w[0] + w[1]) + (w[2] + w[3]);
        }
#pragma unroll
        for (int jt = 0; jt < D; ++jt) {
#pragma unroll
            for (int ib = 0; ib < K; ++ib) w[ib] = tc[ib];
#pragma unroll
            for (int j = 0; j < D; ++j)
                if (j != jt) {
#pragma unroll
                    for (int ib = 0; ib < K; ++ib) w[ib] *= pfix[j];
                }
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int ib = 0; ib < K; ++ib) w[ib] *= pim[D + t][ct[t]];
            sf[jt] += (w[0] + w[1]) + (w[2] + w[3]);
        }
    }

    TILE_STAMP(4);  // contraction over this lane's entries
    // ---- combine inside the G-lane group
#pragma unroll
    for (int mask = 1; mask < G; mask <<= 1) {
#pragma unroll
        for (int i = 0; i < K; ++i) pp[i] += shfl_xor_d(pp[i], mask);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int ct = 0; ct < K; ++ct) ol[t][ct] += shfl_xor_d(ol[t][ct], mask);
    }
    // leading parent jt: sum over the lanes that share digit jt (all other lane digits), then
    // collect the four buckets from the lanes whose other digits are zero
    double of[D][K];
#pragma unroll
    for (int jt = 0; jt < D; ++jt) {
        double x = sf[jt];
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j != jt) {
                x += shfl_xor_d(x, 1 << (2 * (D - 1 - j)));
                x += shfl_xor_d(x, 2 << (2 * (D - 1 - j)));
            }
#pragma unroll
        for (int ct = 0; ct < K; ++ct) of[jt][ct] = shfl_d(x, nl * G + (ct << (2 * (D - 1 - jt))));
    }

    TILE_STAMP(5);  // combined inside the group
    // ---- lanes 0..M of the group finish the node: normalise, residual, stores
    {
        double o[K];
#pragma unroll
        for (int ct = 0; ct < K; ++ct) {
            o[ct] = pp[ct];
#pragma unroll
            for (int jt = 0; jt < M; ++jt)
                if (g == jt + 1) o[ct] = jt < D ? of[jt < D ? jt : 0][ct] : ol[jt >= D ? jt - D : 0][ct];
        }
        if (active && g <= M) {
            normalize_k<K>(o);
            if (g == 0) {
                double2_t y0, y1;
                y0.x = frozen ? piv[0] : o[0]; y0.y = frozen ? piv[1] : o[1];
                y1.x = frozen ? piv[2] : o[2]; y1.y = frozen ? piv[3] : o[3];
                bn_store<NT>(&nout[0 * NPT], y0);
                bn_store<NT>(&nout[1 * NPT], y1);
            } else {
#pragma unroll
                for (int i = 0; i < K; ++i) wres = res_acc(wres, fabs(o[i] - fold[i]));
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    double2_t y;
                    y.x = o[2 * h]; y.y = o[2 * h + 1];
                    bn_store<NT>(&rec_out2[fin.lam + h * fin.stride], y);
                }
            }
        }
    }
    TILE_STAMP(6);  // normalised and stored
#ifdef BN_TILE_CLOCK
    if (lane == 0 && td.slot_base < kTileClockTiles)
        g_tile_clock[td.slot_base][10] = (unsigned long long)G | ((unsigned long long)M << 8) | (4ull << 16) | ((unsigned long long)td.cmax << 24) | (0xffffull << 32);
#endif
    if (active && g == 0) {
        if (io.first && !frozen) {  // initial state (:38-41); a group node always has parents
#pragma unroll
            for (int i = 0; i < K; ++i) { piv[i] = 1.0; lav[i] = 1.0; }
        }
        if (!par_fast) wres = res_acc(wres, parent_role_any<K, NT, 12>(b, io, td, nl, frozen, piv, lav));
    }
    return wres;
}

// ---------------------------------------------------------------------------------------------
// flat tile: ANY arities, a group of G = 8..64 lanes per node (NPT = 64 / G nodes per wavefront).
// Entry e of the reference's row-major CPT (parent assignment slowest, own state fastest) sits in
// lane e % G of the node's group, slot e / G.  Vectors
// live spread over the lanes: lane x of `pim` / `out` is element x of the in-edge messages
// concatenated in parent order, lane i of piv / lav / pin is element i of the node vectors.
//   S <= 128 entries: every term is staged in LDS and each accumulator lane adds its own terms in
//     the reference's order (own state outer, assignment inner, :174-200, :240-266) -- bit-identical
//     to the reference wherever the reference is deterministic (<= 2 parents);
//   larger tables: LDS fp64 atomics (sum re-associated, agrees to rounding).  (Tried: the ordered
//     scheme for up to 1024 entries, re-deriving the factors in every pass -- 1.5x slower than the
//     atomics at 243 and 625 entries.)
// Products always keep the reference's ascending parent / child order.
// ---------------------------------------------------------------------------------------------
constexpr int kFlatOrdered = 128;  // tables up to this many entries take the ordered path
constexpr int kFlatCopies = 8;     // atomics path: accumulator copies (lane & 7), fewer same-address conflicts
constexpr int kFlatW = 128 * kFlatCopies;  // doubles per wave: staged terms or the accumulator copies
constexpr int kFlatLK = 256;   // doubles per wave: the children's lambda-messages (parent role)
constexpr int kFlatLds = kFlatW + kFlatLK;

__device__ __forceinline__ double readlane_d(double x, int src) {  // src wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// G lanes per node (NPT = 64 / G nodes per wave): G < 64 only for tables of at most 2 G entries
// whose vectors fit G lanes (the ordered path); every "lane" below is then a lane of the node's group.
// MsgRef -> Loc with the chunk stride by multiplication: magic = ceil(2^20 / h), exact because a tile-resident
// record has lam - pi = h * npt <= 2^11 (bn_plan.cpp)
__device__ __forceinline__ Loc decode_ref_magic(MsgRef r, unsigned magic) {
    Loc l;
    l.has = r.pi >= 0;
    const bool cut = r.lam < 0;
    const int lam = cut ? ~r.lam : r.lam;
    l.pi = l.has ? r.pi : 0;
    l.lam = l.has ? lam : 0;
    l.stride = !l.has ? 0 : (cut ? 1 : int32_t((unsigned(lam - r.pi) * magic) >> 20));
    return l;
}

template <int G, bool NT>
__device__ __forceinline__ double tile_flat(const BpBuffers& b, const IO& io, const TileDesc& td, const ClassDesc& cg,
                                            int wlane, double* lds) {
    constexpr int MM = kFlatMaxParents;
    constexpr int NPT = kWave / G;
    const int nl = wlane / G, lane = wlane % G, gb = nl * G;  // node of the tile, lane inside its group, first lane
    const bool active = nl < td.n_nodes;
    const int nlc = active ? nl : 0;  // idle groups shadow node 0 (they take part in the shuffles, store nothing)
#ifdef BN_TILE_CLOCK
    const int wlane_for_stamp = wlane;
#endif
    TILE_STAMP(0);
    double* W = lds + nl * (kFlatW / NPT);  // the group's share: kFlatW / NPT = 16 G doubles
    double* LK = lds + kFlatW + nl * (kFlatLK / NPT);
    // the class fields used below, copied once into (scalar) registers: the wave-scope fences
    // between the LDS phases would otherwise make every later use a fresh load
    struct {
        int kv, m, rows, kvp, per_lane, tab;
        unsigned magic_kv, magic_hv;
        int kp[MM], kpp[MM], rec_off[MM], lam_run[MM];
    } c;
    c.kv = cg.kv; c.m = cg.m; c.rows = cg.rows; c.kvp = cg.kvp; c.per_lane = cg.per_lane; c.tab = cg.flat_tab_off;
    c.magic_kv = unsigned(cg.magic_kv); c.magic_hv = unsigned(cg.magic_hv);
#pragma unroll
    for (int j = 0; j < MM; ++j) { c.kp[j] = cg.kp[j]; c.kpp[j] = cg.kpp[j]; c.rec_off[j] = cg.rec_off[j]; c.lam_run[j] = cg.lam_run[j]; }
    const int kv = c.kv, m = c.m, rows = c.rows, kvp = c.kvp;
    const int S = kv * rows;
    auto div_kv = [&](int x) { return int((unsigned(x) * c.magic_kv) >> 16); };  // x / kv for 0 <= x < 1024
    const bool ordered = G < kWave || S <= kFlatOrdered;
    TILE_STAMP(1);  // class descriptor in registers
    double wres = 0.0;

    // ---- every load whose address depends on the descriptors only goes out first: CPT values, the entry
    // table, the in-edge records, marks and node vectors, the out-edge references
    const double* cp = b.cpt + td.cpt_base + (nlc * G + lane) * 2;
    double cval[2] = {0.0, 0.0};
    uint4 fe_lo[2], fe_hi[2];
    if (ordered) {
        const uint4* tab = reinterpret_cast<const uint4*>(b.flat_tab + c.tab);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fe_lo[t] = tab[(lane + G * t) * 2];
            fe_hi[t] = tab[(lane + G * t) * 2 + 1];
            cval[t] = cp[t];  // slots beyond the table are zero-filled by the planner's image (valid decides)
        }
    }
    // in-edge records: lane x <-> (parent j, state d), x = offs[j] + d
    int offs[MM + 1];
    int myj = -1, mykj = 0, myoff = 0, myrun = 0;  // this lane's parent, its arity, first lane and bucket run length
    int64_t my_pi = 0, my_lam = 0;  // doubles from the start of a record buffer
    {
        // the in-edge of this lane is picked with selects inside the loop (everything per j is wave-uniform);
        // the 64-bit element address is formed once, after it
        int64_t sel_pi = 0, sel_lam = 0;
        int sel_stride = 0;
        int off = 0;
#pragma unroll
        for (int j = 0; j < MM; ++j) {
            offs[j] = off;
            if (j < m) {
                const int kj = c.kp[j], hj = c.kpp[j] / 2;
                Loc l;
                if (td.in_ref_base >= 0) {
                    l = decode_ref(b.in_refs[td.in_ref_base + j * NPT + nlc], hj);
                } else {
                    l.has = true; l.pi = (td.rec_base + c.rec_off[j]) / 2 + nlc; l.lam = l.pi + hj * NPT; l.stride = NPT;
                }
                if (lane >= off && lane < off + kj) {
                    myj = j; mykj = kj; myoff = off; myrun = c.lam_run[j];
                    sel_pi = l.pi; sel_lam = l.lam; sel_stride = l.stride;
                }
                off += kj;
            }
        }
        offs[MM] = off;
        const int dd = lane - myoff;
        const int64_t step = int64_t((dd >> 1) * sel_stride);  // chunk offset: < 2^31 (stride <= 64 or 1, dd < 64)
        my_pi = (sel_pi + step) * 2 + (dd & 1);
        my_lam = (sel_lam + step) * 2 + (dd & 1);
    }
    const int sumk = offs[MM];
    double pim = 1.0, oldlam = 1.0;
    if (myj >= 0 && !io.first) { pim = io.rec_in[my_pi]; oldlam = io.rec_in[my_lam]; }
    const bool frozen = b.frozen[td.slot_base + nlc] == b.frozen_mark;
    // node vectors (old), lane i < kv
    const double* nin = io.node_in + td.node_base;
    double* nout = io.node_out + td.node_base;
    // element i of pi(v) / lambda(v) in the tile's striped node block
    auto nidx = [&](int half, int i) { return int64_t(half + (i >> 1)) * (NPT * 2) + nlc * 2 + (i & 1); };
    // requested whether or not they will be used (deciding first would put the mark's round trip in front of them);
    // iteration 0 of a node without evidence starts from 1.0 / its CPT row instead (:33-73)
    double piv = 1.0, lav = 1.0;
    if (lane < kv) {
        const double old_pi = nin[nidx(0, lane)], old_lam = nin[nidx(kvp / 2, lane)];
        const double root = m == 0 ? b.cpt[td.cpt_base + (nlc * G + lane) * 2] : 1.0;  // a root starts from its CPT row (:58-64)
        const bool kept = !io.first || frozen;
        piv = kept ? old_pi : root;
        lav = kept ? old_lam : 1.0;
    }

    // ---- parent role, first pass: child (c, i) <-> lane c*kv + i.  Its reference heads a dependent chain
    // (reference -> the child's lambda-message element and the previous pi-message element): the reference is
    // requested now with everything else, the second trip once the child role's terms are staged, so that it
    // is in flight while they are summed and normalised.
    const int cmax = td.cmax;
    const int ptotal = cmax * kv;
    const int pchunk = div_kv(G) * kv;  // whole children per pass
    const bool pstaged = ptotal <= kFlatLK / NPT;
    const bool pfirst = pstaged && lane < pchunk && lane < ptotal;
    int pi0 = 0;
    MsgRef pr0{-1, 0};
    if (pfirst) {
        const int pc0 = div_kv(lane);
        pi0 = lane - pc0 * kv;
        pr0 = b.out_refs[td.out_base + int64_t(pc0) * NPT + nlc];
    }
    Loc pl0;
    pl0.has = false; pl0.pi = 0; pl0.lam = 0; pl0.stride = 0;
    double plk0 = 1.0, pold0 = 1.0;  // a missing child contributes 1.0 (x * 1.0 == x)
    auto parent_second_trip = [&]() {
        if (pfirst) {
            pl0 = decode_ref_magic(pr0, c.magic_hv);
            if (pl0.has && !io.first) {
                plk0 = io.rec_in[(pl0.lam + int64_t(pi0 >> 1) * pl0.stride) * 2 + (pi0 & 1)];
                pold0 = io.rec_in[(pl0.pi + int64_t(pi0 >> 1) * pl0.stride) * 2 + (pi0 & 1)];
            }
        }
    };
    TILE_STAMP(2);  // first-trip loads issued

    double outl = 0.0;  // lane x: un-normalised lambda-message element x (concatenated)
    double pin = 0.0;   // lane i: un-normalised pi(v)[i]

    if (ordered) {
        // ---- ordered path: at most two entries per lane.  EVERY term -- of pi(v) and of the lambda-message to
        // each parent -- is written to LDS at the position it has in ITS accumulator's summation order (FlatEntry,
        // computed by the planner), region r of the group's share holding accumulator set r (0: pi(v), 1 + jt:
        // parent jt); one fence; then each accumulator lane adds its contiguous run front to back (additions
        // strictly in the reference's order :174-200, :240-266).
        constexpr int kRegion = 2 * G;                 // doubles per region (S <= 2 G)
        constexpr int kRegions = (kFlatW / NPT) / kRegion;  // 8
        double pj[2][MM], li[2], cv[2];
        bool ok[2];
        unsigned ei[2], pos_pi[2], dj[2][MM], pos_lam[2][MM];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned w0[4] = {fe_lo[t].x, fe_lo[t].y, fe_lo[t].z, fe_lo[t].w};
#pragma unroll
            for (int j = 0; j < MM; ++j) {
                dj[t][j] = (w0[j >> 2] >> (8 * (j & 3))) & 255u;
                pos_lam[t][j] = (w0[2 + (j >> 2)] >> (8 * (j & 3))) & 255u;
            }
            ei[t] = fe_hi[t].x & 255u;
            pos_pi[t] = (fe_hi[t].x >> 8) & 255u;
            ok[t] = ((fe_hi[t].x >> 16) & 255u) != 0;
#pragma unroll
            for (int j = 0; j < MM; ++j) {
                pj[t][j] = 1.0;
                if (j < m) pj[t][j] = shfl_d(pim, gb + offs[j] + int(dj[t][j]));
            }
            li[t] = shfl_d(lav, gb + int(ei[t]));
            cv[t] = ok[t] ? cval[t] : 0.0;
        }
        TILE_STAMP(3);  // in-edge messages and CPT values have arrived
        // calculate_pi (:174-200): cpt * pi-messages (ascending parents)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            double v = cv[t];
#pragma unroll
            for (int j = 0; j < MM; ++j)
                if (j < m) v *= pj[t][j];
            if (ok[t]) W[pos_pi[t]] = v;
        }
        // calculate_lambda_k (:240-266) per target parent: (lambda[i] * cpt) * the OTHER parents' pi-messages
        auto lambda_terms = [&](int jt, int region) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                double w = li[t] * cv[t];
#pragma unroll
                for (int j = 0; j < MM; ++j)
                    if (j < m && j != jt) w *= pj[t][j];
                if (ok[t]) W[region * kRegion + int(pos_lam[t][jt])] = w;
            }
        };
#pragma unroll
        for (int jt = 0; jt < kRegions - 1; ++jt)
            if (jt < m) lambda_terms(jt, 1 + jt);
        wave_lds_fence();
        TILE_STAMP(4);  // terms staged
        parent_second_trip();
        {
            // both runs of a lane (pi(v)[lane] if lane < kv, its lambda-message element if it has one) are summed in one loop
            const int Rpi = lane < kv ? rows : 0;
            const int Rlam = (myj >= 0 && myj < kRegions - 1) ? myrun : 0;
            const double* run_pi = W + (lane < kv ? lane : 0) * rows;
            const double* run_lam = W + (1 + (myj < 0 ? 0 : myj)) * kRegion + (lane - myoff) * myrun;
            int Rtop = rows;
#pragma unroll
            for (int j = 0; j < MM; ++j)
                if (j < m) Rtop = c.lam_run[j] > Rtop ? c.lam_run[j] : Rtop;
            int Rmin = rows;
#pragma unroll
            for (int j = 0; j < MM; ++j)
                if (j < m) Rmin = c.lam_run[j] < Rmin ? c.lam_run[j] : Rmin;
            // Eight consecutive terms of each run per step, read unconditionally (a read past a run's end stays
            // inside this wave's LDS slice: the staging area is followed by the children's area): no remainder
            // loop, no branches, reads with immediate offsets.  While every run of the tile has eight terms left
            // the step is loads + additions only; after that a term past its run's end becomes +0.0, which leaves
            // the sum as it is (a sum started from +0.0 is never -0.0), the selects off the chain of additions.
            constexpr int kStep = 8;
            double acc_pi = 0.0, acc_lam = 0.0;
            int r0 = 0;
            for (; r0 + kStep <= Rmin; r0 += kStep) {
                double x_pi[kStep], x_lam[kStep];
#pragma unroll
                for (int q = 0; q < kStep; ++q) {
                    x_pi[q] = run_pi[r0 + q];
                    x_lam[q] = run_lam[r0 + q];
                }
#pragma unroll
                for (int q = 0; q < kStep; ++q) {
                    acc_pi += x_pi[q];
                    acc_lam += x_lam[q];
                }
            }
            for (; r0 < Rtop; r0 += kStep) {
                double x_pi[kStep], x_lam[kStep];
#pragma unroll
                for (int q = 0; q < kStep; ++q) {
                    x_pi[q] = run_pi[r0 + q];
                    x_lam[q] = run_lam[r0 + q];
                }
#pragma unroll
                for (int q = 0; q < kStep; ++q) {
                    x_pi[q] = r0 + q < Rpi ? x_pi[q] : 0.0;
                    x_lam[q] = r0 + q < Rlam ? x_lam[q] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < kStep; ++q) {
                    acc_pi += x_pi[q];
                    acc_lam += x_lam[q];
                }
            }
            pin = acc_pi;
            outl = acc_lam;
        }
        if (m > kRegions - 1) {  // an eighth parent: its terms reuse region 0 once pi(v) has been summed
            wave_lds_fence();
            lambda_terms(MM - 1, 0);
            wave_lds_fence();
            if (myj == MM - 1) {
                const double* run = W + (lane - myoff) * myrun;
                double acc = 0.0;
#pragma unroll 8
                for (int r = 0; r < myrun; ++r) acc += run[r];
                outl = acc;
            }
        }
        wave_lds_fence();
    } else {
        // ---- large table: LDS atomics into [ lambda buckets (sumk) | pi (kv) ]
        parent_second_trip();
        const int nacc = sumk + kv;  // <= 128
        for (int x = lane; x < nacc * kFlatCopies; x += kWave) W[x] = 0.0;
        wave_lds_fence();
        double* Wc = W + (lane & (kFlatCopies - 1)) * nacc;  // this lane's accumulator copy
        const int T = c.per_lane;
        double cvals[4] = {0.0, 0.0, 0.0, 0.0};
        // digits of this lane's entry, advanced by 64 per step with carries (one set of divisions
        // per tile instead of one per entry): own state fastest, last parent next, first parent slowest
        int inc_i, inc_j[MM], cur_i, cur_j[MM];
        {
            int q = kWave, e0 = lane;
            inc_i = q % kv; q /= kv;
            cur_i = e0 % kv; e0 /= kv;
#pragma unroll
            for (int j = MM - 1; j >= 0; --j) {
                inc_j[j] = 0; cur_j[j] = 0;
                if (j < m) {
                    inc_j[j] = q % c.kp[j]; q /= c.kp[j];
                    cur_j[j] = e0 % c.kp[j]; e0 /= c.kp[j];
                }
            }
        }
        for (int t = 0; t < T; ++t) {
            if ((t & 3) == 0) {  // the CPT values of four entries are requested together
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tq = t + q;
                    cvals[q] = (tq < T && lane + kWave * tq < S) ? cp[int64_t(tq >> 1) * 128 + (tq & 1)] : 0.0;
                }
            }
            const int e = lane + kWave * t;
            const bool valid = e < S;
            const double cvq = (t & 3) == 0 ? cvals[0] : ((t & 3) == 1 ? cvals[1] : ((t & 3) == 2 ? cvals[2] : cvals[3]));
            double pj[MM], li, cv;
            int dj[MM], ei;
            ei = valid ? cur_i : 0;
#pragma unroll
            for (int j = 0; j < MM; ++j) {
                dj[j] = valid ? cur_j[j] : 0;
                pj[j] = 1.0;
                if (j < m) pj[j] = shfl_d(pim, gb + offs[j] + dj[j]);
            }
            li = shfl_d(lav, gb + ei);
            cv = valid ? cvq : 0.0;
            {   // e += 64 in mixed radix
                int x = cur_i + inc_i;
                int carry = x >= kv ? 1 : 0;
                cur_i = x - (carry ? kv : 0);
#pragma unroll
                for (int j = MM - 1; j >= 0; --j)
                    if (j < m) {
                        x = cur_j[j] + inc_j[j] + carry;
                        carry = x >= c.kp[j] ? 1 : 0;
                        cur_j[j] = x - (carry ? c.kp[j] : 0);
                    }
            }
            double v = cv;
#pragma unroll
            for (int j = 0; j < MM; ++j)
                if (j < m) v *= pj[j];
            if (valid) unsafeAtomicAdd(&Wc[sumk + ei], v);
            const double tc = li * cv;
#pragma unroll
            for (int jt = 0; jt < MM; ++jt) {
                if (jt < m) {
                    double w = tc;
#pragma unroll
                    for (int j = 0; j < MM; ++j)
                        if (j < m && j != jt) w *= pj[j];
                    if (valid) unsafeAtomicAdd(&Wc[offs[jt] + dj[jt]], w);
                }
            }
        }
        wave_lds_fence();
        if (lane < sumk) {
            double acc = W[lane];
#pragma unroll
            for (int q = 1; q < kFlatCopies; ++q) acc += W[q * nacc + lane];
            outl = acc;
        }
        if (lane < kv) {
            double acc = W[sumk + lane];
#pragma unroll
            for (int q = 1; q < kFlatCopies; ++q) acc += W[q * nacc + sumk + lane];
            pin = acc;
        }
        wave_lds_fence();
    }
    TILE_STAMP(5);  // pi(v) and the lambda-messages summed

    // ---- normalise (:298-311: divide by the plain left-to-right sum), residual (:105-131), stores.
    // The un-normalised elements go through LDS once; every lane then adds the elements of ITS vector front to
    // back (the lanes of a vector compute the same sum), instead of one broadcast per element in turn.
    {
        if (lane < sumk) W[lane] = outl;
        if (lane < kv) W[G + lane] = pin;
        wave_lds_fence();
        const int n_pi = lane < kv ? kv : 0;
        const int n_lam = mykj;  // 0 for a lane without an in-edge element
        const double* v_pi = W + G;
        const double* v_lam = W + myoff;
        int top = kv;
#pragma unroll
        for (int j = 0; j < MM; ++j)
            if (j < m) top = c.kp[j] > top ? c.kp[j] : top;
        const int l_pi = n_pi > 0 ? n_pi - 1 : 0, l_lam = n_lam > 0 ? n_lam - 1 : 0;
        double sum_pi = 0.0, sum_lam = 0.0;
        for (int r0 = 0; r0 < top; r0 += 4) {
            double x_pi[4], x_lam[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                x_pi[q] = v_pi[r0 + q < l_pi ? r0 + q : l_pi];
                x_lam[q] = v_lam[r0 + q < l_lam ? r0 + q : l_lam];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // + 0.0 past the end (see above)
                sum_pi += r0 + q < n_pi ? x_pi[q] : 0.0;
                sum_lam += r0 + q < n_lam ? x_lam[q] : 0.0;
            }
        }
        wave_lds_fence();
        pin /= sum_pi;
        if (myj >= 0) outl /= sum_lam;
        if (active && lane < kv) nout[nidx(0, lane)] = frozen ? piv : pin;
        if (active && lane == kv && kvp > kv) nout[nidx(0, lane)] = 0.0;
    }
    if (active && myj >= 0) {
        wres = res_acc(wres, fabs(outl - oldlam));
        io.rec_out[my_lam] = outl;
    }
    TILE_STAMP(6);  // normalised, child-role stores issued

    // ---- parent role (:202-238): the children's lambda-messages staged in LDS, element (c, i) at c*kv + i
    if (pstaged) {
        const int total = ptotal, chunk = pchunk;
        if (lane < chunk && lane < total) LK[lane] = plk0;
        for (int base = chunk; base < total; base += chunk) {
            const int idx = base + lane;
            if (lane < chunk && idx < total) {
                const int cc = div_kv(idx), ii = idx - cc * kv;
                const Loc l = decode_ref_magic(b.out_refs[td.out_base + int64_t(cc) * NPT + nlc], c.magic_hv);
                double val = 1.0;
                if (l.has && !io.first) val = io.rec_in[(l.lam + int64_t(ii >> 1) * l.stride) * 2 + (ii & 1)];
                LK[idx] = val;
            }
        }
        wave_lds_fence();
        TILE_STAMP(7);  // children's messages staged
        // product over the children x (ascending, :229-235 / :208-215) of element i of their lambda-messages,
        // skipping child `skip`; four LDS reads in flight per step
        auto children_product = [&](double init, int i, int skip) {
            double acc = init;
            for (int x0 = 0; x0 < cmax; x0 += 4) {
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = LK[(x0 + q < cmax ? x0 + q : cmax - 1) * kv + i];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc *= (x0 + q < cmax && x0 + q != skip) ? v[q] : 1.0;  // x * 1.0 == x
            }
            return acc;
        };
        // left-to-right sum (:298-311) of the kv elements held by lanes first .. first + kv - 1 of the group
        auto vector_sum = [&](double x, int first) {
            double sum = 0.0;
            for (int i0 = 0; i0 < kv; i0 += 4) {
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = shfl_d(x, gb + first + (i0 + q < kv ? i0 + q : 0));
#pragma unroll
                for (int q = 0; q < 4; ++q) sum += i0 + q < kv ? v[q] : 0.0;
            }
            return sum;
        };
        {   // lambda(v): product of the children's lambda-messages from 1.0, ascending (:220-238)
            double acc = children_product(1.0, lane < kv ? lane : 0, -1);
            acc /= vector_sum(acc, 0);
            if (active && lane < kv) nout[nidx(kvp / 2, lane)] = frozen ? lav : acc;
            if (active && lane == kv && kvp > kv) nout[nidx(kvp / 2, lane)] = 0.0;
        }
        for (int base = 0; base < total; base += chunk) {
            const int idx = base + lane;
            const bool mine = lane < chunk && idx < total;
            const int cc = mine ? div_kv(idx) : 0, ii = mine ? idx - cc * kv : 0;
            // pi-message to child cc (:202-218): pi(v)[i] * the OTHER children's lambda-messages, ascending
            double u = children_product(shfl_d(piv, gb + ii), ii, cc);
            u /= vector_sum(u, lane - ii);
            if (mine && active) {
                const Loc l = base == 0 ? pl0 : decode_ref_magic(b.out_refs[td.out_base + int64_t(cc) * NPT + nlc], c.magic_hv);
                if (l.has) {
                    const int64_t at = (l.pi + int64_t(ii >> 1) * l.stride) * 2 + (ii & 1);
                    const double old = io.first ? 1.0 : (base == 0 ? pold0 : io.rec_in[at]);
                    wres = res_acc(wres, fabs(u - old));
                    io.rec_out[at] = u;
                }
            }
        }
        TILE_STAMP(8);  // parent role done
#ifdef BN_TILE_CLOCK
        if (wlane == 0 && td.slot_base < kTileClockTiles) {
            g_tile_clock[td.slot_base][10] = (unsigned long long)G | ((unsigned long long)m << 8) | ((unsigned long long)kv << 16) |
                                             ((unsigned long long)cmax << 24) | ((unsigned long long)rows << 32);
        }
#endif
    } else if (lane == 0 && active) {
        wres = res_acc(wres, parent_role_generic(b, io, td, kv, kvp, nl, frozen, true));
    }
    return wres;
}

template <int K, int M, bool NT>
__device__ __forceinline__ double tile_uniform_dispatch(const BpBuffers& b, const IO& io, const TileDesc& td, int lane) {
    if (td.in_ref_base >= 0) return tile_uniform<K, M, 0, NT, true>(b, io, td, lane);  // boundary tile
    if (td.cmax <= 2) return tile_uniform<K, M, 2, NT, false>(b, io, td, lane);
    if (td.cmax <= 4) return tile_uniform<K, M, 4, NT, false>(b, io, td, lane);
    return tile_uniform<K, M, 0, NT, false>(b, io, td, lane);
}


constexpr int kVarU = 1 << kVariantUniform;
constexpr int kVarUG = kVarU | (1 << kVariantGroup);
constexpr int kVarAll = kVarUG | (1 << kVariantFlat);  // + the one-lane fallback

// One tile, whatever its variant.  VARIANTS (bit v = variant v may occur) compiles the others out:
// the any-arity variant is the only one that needs LDS, the lane-group variant the one that needs the
// most registers; a network of register-resident tiles only runs an instantiation without either.
template <bool NT, int VARIANTS>
__device__ __forceinline__ double run_tile(const BpBuffers& b, const IO& io, const TileDesc& td, int lane, double* lds) {
    constexpr bool FLAT = (VARIANTS >> kVariantFlat) & 1, GROUP = (VARIANTS >> kVariantGroup) & 1;
    double wres = 0.0;
    bool handled = false;
    if (td.variant == kVariantUniform) {
        handled = true;
        switch (td.kv * 8 + td.m) {
            case 2 * 8 + 0: wres = tile_uniform_dispatch<2, 0, NT>(b, io, td, lane); break;
            case 2 * 8 + 1: wres = tile_uniform_dispatch<2, 1, NT>(b, io, td, lane); break;
            case 2 * 8 + 2: wres = tile_uniform_dispatch<2, 2, NT>(b, io, td, lane); break;
            case 2 * 8 + 3: wres = tile_uniform_dispatch<2, 3, NT>(b, io, td, lane); break;
            case 2 * 8 + 4: wres = tile_uniform_dispatch<2, 4, NT>(b, io, td, lane); break;
            case 3 * 8 + 0: wres = tile_uniform_dispatch<3, 0, NT>(b, io, td, lane); break;
            case 3 * 8 + 1: wres = tile_uniform_dispatch<3, 1, NT>(b, io, td, lane); break;
            case 3 * 8 + 2: wres = tile_uniform_dispatch<3, 2, NT>(b, io, td, lane); break;
            case 4 * 8 + 0: wres = tile_uniform_dispatch<4, 0, NT>(b, io, td, lane); break;
            case 4 * 8 + 1: wres = tile_uniform_dispatch<4, 1, NT>(b, io, td, lane); break;
            case 4 * 8 + 2: wres = tile_uniform_dispatch<4, 2, NT>(b, io, td, lane); break;
            default: handled = false; break;
        }
    }
    if constexpr (GROUP) {
        if (td.variant == kVariantGroup) {
            handled = true;
            switch (td.m) {
                // npt tells the split: 4^(m-2) lanes per node (64 entries per lane) or 4^(m-1) (16 entries per lane)
                case 3: wres = td.npt == 16 ? tile_group<1, 2, NT>(b, io, td, lane) : tile_group<2, 1, NT>(b, io, td, lane); break;
                case 4: wres = td.npt == 4 ? tile_group<2, 2, NT>(b, io, td, lane) : tile_group<3, 1, NT>(b, io, td, lane); break;
                case 5: wres = tile_group<3, 2, NT>(b, io, td, lane); break;
                default: handled = false; break;
            }
        }
    }
    if constexpr (FLAT) {
        if (td.variant == kVariantFlat) {
            handled = true;
            switch (td.npt) {  // 64 / npt lanes per node
                case 1: wres = tile_flat<64, NT>(b, io, td, b.classes[td.cls], lane, lds); break;
                case 2: wres = tile_flat<32, NT>(b, io, td, b.classes[td.cls], lane, lds); break;
                case 4: wres = tile_flat<16, NT>(b, io, td, b.classes[td.cls], lane, lds); break;
                default: wres = tile_flat<8, NT>(b, io, td, b.classes[td.cls], lane, lds); break;
            }
        }
    }
    if constexpr (FLAT) {  // the one-lane fallback exists in the all-variants instantiation only
        if (!handled) wres = tile_generic(b, io, td, b.classes[td.cls], lane);
    }
    return wres;
}

// the any-arity and one-lane variants only (the high-occupancy instantiations)
__device__ __forceinline__ double run_tile_light(const BpBuffers& b, const IO& io, const TileDesc& td, int lane, double* lds) {
    if (td.variant == kVariantFlat) {
        switch (td.npt) {  // 64 / npt lanes per node
            case 1: return tile_flat<64, false>(b, io, td, b.classes[td.cls], lane, lds);
            case 2: return tile_flat<32, false>(b, io, td, b.classes[td.cls], lane, lds);
            case 4: return tile_flat<16, false>(b, io, td, b.classes[td.cls], lane, lds);
            default: return tile_flat<8, false>(b, io, td, b.classes[td.cls], lane, lds);
        }
    }
    return tile_generic(b, io, td, b.classes[td.cls], lane);
}

// belief = normalize(pi % lambda) (:151-158) of one tile's nodes from node buffer `node_buf`.
// Lane nl serves node nl: its pi / lambda chunks are 16-byte loads, coalesced across the lanes (chunk h of
// node nl sits at double2 index h * npt + nl); the k products stay in registers, one left-to-right sum,
// one division each; the belief goes out in 16-byte stores where the node's offset allows (even sum of
// the preceding arities), nothing is read back.  H = chunks per vector, 1..4 (k <= 8) unrolled.
template <int H>
__device__ __forceinline__ void tile_beliefs_regs(const BpBuffers& b, const TileDesc& td, const double* node_buf, int lane) {
    const int npt = td.npt, kv = td.kv;
    const double2_t* node = reinterpret_cast<const double2_t*>(node_buf + td.node_base) + lane;
    double2_t p[H], l[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        p[h] = node[h * npt];
        l[h] = node[(H + h) * npt];
    }
    const int64_t boff = b.slot_boff[td.slot_base + lane];
    double x[2 * H];
    double sum = 0;
#pragma unroll
    for (int h = 0; h < H; ++h) {
        x[2 * h] = p[h].x * l[h].x;
        x[2 * h + 1] = p[h].y * l[h].y;
    }
#pragma unroll
    for (int i = 0; i < 2 * H; ++i)
        if (i < kv) sum += x[i];
#pragma unroll
    for (int i = 0; i < 2 * H; ++i) x[i] = x[i] / sum;
    double* out = b.beliefs + boff;
    if ((boff & 1) == 0) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            if (2 * h + 1 < kv) {
                double2_t y;
                y.x = x[2 * h]; y.y = x[2 * h + 1];
                reinterpret_cast<double2_t*>(out)[h] = y;
            } else if (2 * h < kv) {
                out[2 * h] = x[2 * h];
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2 * H; ++i)
            if (i < kv) out[i] = x[i];
    }
}

__device__ __forceinline__ void tile_beliefs(const BpBuffers& b, const TileDesc& td, const double* node_buf, int lane) {
    if (lane >= td.n_nodes) return;  // one lane per node writes the belief
    const int half = ((td.kv + 1) & ~1) >> 1;
    switch (half) {
        case 1: return tile_beliefs_regs<1>(b, td, node_buf, lane);
        case 2: return tile_beliefs_regs<2>(b, td, node_buf, lane);
        case 3: return tile_beliefs_regs<3>(b, td, node_buf, lane);
        case 4: return tile_beliefs_regs<4>(b, td, node_buf, lane);
        default: break;
    }
    // arity above 8: two passes over the (cached) inputs, nothing read back from the output
    const double* node = node_buf + td.node_base;
    const int64_t boff = b.slot_boff[td.slot_base + lane];
    double sum = 0;
    for (int i = 0; i < td.kv; ++i) sum += node[vidx(0, i, td.npt, lane)] * node[vidx(half, i, td.npt, lane)];
    for (int i = 0; i < td.kv; ++i)
        b.beliefs[boff + i] = (node[vidx(0, i, td.npt, lane)] * node[vidx(half, i, td.npt, lane)]) / sum;
}

}  // namespace bnmi

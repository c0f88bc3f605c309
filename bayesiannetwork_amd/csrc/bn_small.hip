// bn_small.hip -- the whole belief-propagation run of a SMALL network in ONE workgroup, state in LDS.
// Work items, encodings and the reasoning: bn_small.hpp / bn_small_plan.cpp.  Reference: belief_propagation.hpp:33-158.
//
// One iteration of the reference's while(true) loop (:75-148) =
//   phase 1   entry items: the terms of pi(v) (:174-200) and of the lambda-messages (:240-266) -> staging;
//   barrier
//   phase 2   accumulator items: each adds its run of staged terms front to back, normalises (:298-311), stores;
//             product items: lambda(v) (:220-238) and the pi-messages (:202-218) from the OLD state, normalised, stored;
//             maximum_difference (:105-131) of the wave -> LDS;
//   barrier   every wave reads the same words and takes the same stop decision (:147, strict <).
// Old and new state are the two halves of double buffers (the reference's new_* maps, :135-143).
#include "bn_small.hpp"
#include "bn_tiles.hpp"
#include "bn_small_dev.hpp"

namespace bnmi {

// diagnostic builds (make EXTRA=-DBN_TILE_CLOCK): lane 0 of every wave records the 100 MHz clock at the phase
// boundaries of iteration 3 (scripts/experiments/small_clock.py prints them)
#ifdef BN_TILE_CLOCK
__device__ unsigned long long g_small_clock[kSmallMaxWaves][8];
#define SMALL_STAMP(k) do { if (lane == 0 && s == a.sweep_begin + 3) g_small_clock[wave][k] = wall_clock64(); } while (0)
#else
#define SMALL_STAMP(k) do { } while (0)
#endif

struct SmallLds {
    double* pi;     // [2][M]  (buffer c of an array at + c * its length: no pointer tables, they would live in scratch memory)
    double* lam;    // [2][M]
    double* npi;    // [2][N]
    double* nlam;   // [2][N]
    double* stg;
    uint32_t* term;
    uint16_t* clist;
    uint8_t* frz;
    unsigned long long* red;
};

__device__ __forceinline__ SmallLds small_carve(char* base, const SmallArgs& a) {
    SmallLds L;
    double* d = reinterpret_cast<double*>(base);
    L.pi = d; d += 2 * a.M;
    L.lam = d; d += 2 * a.M;
    L.npi = d; d += 2 * a.N;
    L.nlam = d; d += 2 * a.N;
    L.stg = d; d += a.T;
    L.term = reinterpret_cast<uint32_t*>(d);
    char* c = reinterpret_cast<char*>(L.term + (((a.TT > 0 ? a.TT : 1) + 1) & ~1));  // the arrays behind stay 8-byte aligned
    L.clist = reinterpret_cast<uint16_t*>(c);
    c += (size_t(a.CL > 0 ? a.CL : 1) * 2 + 7) & ~size_t(7);
    L.frz = reinterpret_cast<uint8_t*>(c);
    c += (size_t(a.N) + 7) & ~size_t(7);
    L.red = reinterpret_cast<unsigned long long*>(c);
    return L;
}

// One CPT entry: cpt * pi-messages in ascending parent order (:174-200) and, per target parent, (lambda(v)[i] * cpt)
// * the OTHER parents' pi-messages in ascending order (:240-266).  MM = the wave's largest parent count: the loops are
// unrolled to it, a lane with fewer parents multiplies by 1.0 (x * 1.0 == x).
template <int MM, bool REG>
__device__ __forceinline__ void small_entry(const SmallLds& L, const double* pi_cur, const double* nlam_cur, SmallEntry h, double c,
                                            const uint32_t (&treg)[4]) {
    if (((h.y >> 24) & 1u) == 0) return;
    const int m = int((h.y >> 16) & 0xffu), tbase = int(h.y & 0xffffu);
    const double li = nlam_cur[h.x & 0xffffu];
    uint32_t tw[MM > 0 ? MM : 1];
    double pj[MM > 0 ? MM : 1];
#pragma unroll
    for (int j = 0; j < MM; ++j) {
        if (REG && MM <= 4) tw[j] = treg[j < 4 ? j : 0];  // (an entry's parent terms never change: kept in registers when they fit)
        else tw[j] = j < m ? L.term[tbase + j] : 0u;
    }
#pragma unroll
    for (int j = 0; j < MM; ++j) {
        const double x = pi_cur[tw[j] & 0xffffu];
        pj[j] = j < m ? x : 1.0;
    }
    double v = c;
#pragma unroll
    for (int j = 0; j < MM; ++j) v *= pj[j];
    L.stg[h.x >> 16] = v;
    const double lc = li * c;
#pragma unroll
    for (int jt = 0; jt < MM; ++jt) {
        double w = lc;
#pragma unroll
        for (int j = 0; j < MM; ++j)
            if (j != jt) w *= pj[j];
        if (jt < m) L.stg[tw[jt] >> 16] = w;
    }
}

template <bool REG>
__device__ __forceinline__ void small_entry_any(int mm, const SmallLds& L, const double* pi_cur, const double* nlam_cur, SmallEntry h, double c,
                                                const uint32_t (&treg)[4]) {
    switch (mm) {
        case 0: return small_entry<0, REG>(L, pi_cur, nlam_cur, h, c, treg);
        case 1: return small_entry<1, REG>(L, pi_cur, nlam_cur, h, c, treg);
        case 2: return small_entry<2, REG>(L, pi_cur, nlam_cur, h, c, treg);
        case 3: return small_entry<3, REG>(L, pi_cur, nlam_cur, h, c, treg);
        case 4: return small_entry<4, REG>(L, pi_cur, nlam_cur, h, c, treg);
        case 5: case 6: return small_entry<6, REG>(L, pi_cur, nlam_cur, h, c, treg);
        default: return small_entry<8, REG>(L, pi_cur, nlam_cur, h, c, treg);
    }
}

// Normalisation of the vector whose elements sit in adjacent lanes of this wave: the un-normalised element goes to
// its place in `buf`, every lane then adds the vector's elements front to back (:298-311: plain left-to-right sum)
// and divides.  kmax = the wave's largest arity.  LDS operations of one wave execute in order: the caller's final
// store to the same place comes after every lane's reads.
__device__ __forceinline__ double small_normalize(double* buf, bool on, int out_idx, int k, int at, double val, int kmax) {
    if (on) buf[out_idx] = val;
    lds_fence();
    const int vec = on ? out_idx - at : 0;
    double sum = 0.0;
    for (int r0 = 0; r0 < kmax; r0 += 4) {
        double x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = buf[vec + (r0 + q < k ? r0 + q : 0)];
#pragma unroll
        for (int q = 0; q < 4; ++q) sum += r0 + q < k ? x[q] : 0.0;  // + 0.0 past the end: a sum started from +0.0 is never -0.0
    }
    lds_fence();
    return val / sum;
}

// ROUNDS = items of one kind per thread (1, 2 or kSmallMaxRounds): a network that fits one round keeps a third of the registers
template <int ROUNDS>
__global__ __launch_bounds__(kSmallMaxWaves * kWave) void bp_small_kernel(SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) char small_lds[];
    const int set = blockIdx.x;
    BpBuffers b = a.b;
    shift_to_set(b, a.sets, set);
    double* state = a.state + int64_t(set) * a.state_stride;
    const SmallLds L = small_carve(small_lds, a);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1);
    [[maybe_unused]] const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long t_first = wall_clock64();

    // ---- this thread's items, kept in registers for the whole run
    SmallEntry ent[ROUNDS];
    double ecpt[ROUNDS];
    SmallSlot bs[ROUNDS], cs[ROUNDS];
    constexpr bool kTermsInRegs = ROUNDS == 1;
    uint32_t treg[ROUNDS][4];
    int e_mm[ROUNDS], b_rmax[ROUNDS], b_kmax[ROUNDS], c_dmax[ROUNDS], c_kmax[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        ent[r] = SmallEntry{0u, 0u}; ecpt[r] = 0.0;
        bs[r] = SmallSlot{0u, 0u, 0u, 0u}; cs[r] = SmallSlot{0u, 0u, 0u, 0u};
        if (r < a.re) { ent[r] = a.ent[r * nt + tid]; ecpt[r] = a.ent_cpt[r * nt + tid]; }
        if (r < a.rb) bs[r] = a.bslot[r * nt + tid];
        if (r < a.rc) cs[r] = a.cslot[r * nt + tid];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            treg[r][j] = 0u;
            if (kTermsInRegs && ((ent[r].y >> 24) & 1u) && j < int((ent[r].y >> 16) & 0xffu)) treg[r][j] = a.term[(ent[r].y & 0xffffu) + j];
        }
        e_mm[r] = wave_imax(((ent[r].y >> 24) & 1u) ? int((ent[r].y >> 16) & 0xffu) : 0);
        b_rmax[r] = wave_imax(int(bs[r].x >> 16));
        b_kmax[r] = wave_imax(bs[r].z != 0 ? int((bs[r].y >> 16) & 0xffu) : 0);
        c_dmax[r] = wave_imax(int(cs[r].x >> 16));
        c_kmax[r] = wave_imax((cs[r].z & 0xffu) != 0 ? int((cs[r].y >> 16) & 0xffu) : 0);
    }
    // ---- tables and initial state (:33-73) into LDS
    for (int t = tid; t < a.TT; t += nt) L.term[t] = a.term[t];
    for (int t = tid; t < a.CL; t += nt) L.clist[t] = a.clist[t];
    for (int t = tid; t < a.T; t += nt) L.stg[t] = 0.0;  // the padding of the runs stays zero for the whole run
    int s = a.sweep_begin;
    {
        const int c0 = s & 1;
        if (a.ev_mode == 0) {
            for (int y = tid; y < a.N; y += nt) {
                const bool frozen = b.frozen[a.nv_slot[y]] == b.frozen_mark;  // preconditional_node_ (:69)
                L.frz[y] = frozen ? 1 : 0;
                if (s == 0) {
                    const double ev = b.node0[a.nv_idx[y]];  // where bp_evidence_kernel left the evidence vector
                    L.npi[c0 * a.N + y] = frozen ? ev : a.npi_init[y];
                    L.nlam[c0 * a.N + y] = frozen ? ev : 1.0;
                } else {
                    L.npi[c0 * a.N + y] = state[2 * a.M + y];
                    L.nlam[c0 * a.N + y] = state[2 * a.M + a.N + y];
                }
            }
        } else {
            // the evidence arrays themselves (mapped host memory): nodes, offsets AND values are requested together -- the
            // values' range is known from the set's header, so they travel while the initial state is written and wait in
            // the staging array until the offsets say where they belong (one trip over PCIe instead of two dependent ones)
            const int32_t* meta = a.ev_meta ? a.ev_meta + 8 * set : nullptr;
            const int ne = meta ? meta[0] : a.ev_ne, nval = meta ? meta[4] : a.ev_nval;
            const int32_t* ev_node = a.ev_node + (meta ? meta[1] : 0);
            const int32_t* ev_off = a.ev_off + (meta ? meta[2] : 0);
            const double* ev_val = a.ev_val + (meta ? meta[3] : 0);
            int v0 = 0, o0 = 0;
            double x0 = 0.0;
            if (tid < ne) { v0 = ev_node[tid]; o0 = ev_off[tid]; }
            if (tid < nval && s == 0) x0 = ev_val[tid];
            for (int y = tid; y < a.N; y += nt) {
                L.frz[y] = 0;
                L.npi[c0 * a.N + y] = s == 0 ? a.npi_init[y] : state[2 * a.M + y];
                L.nlam[c0 * a.N + y] = s == 0 ? 1.0 : state[2 * a.M + a.N + y];
            }
            __syncthreads();   // (the staging array has been zeroed by every thread's loop above)
            const bool vals_in_lds = nval <= a.T;
            if (s == 0 && vals_in_lds)
                for (int t = tid; t < nval; t += nt) L.stg[t] = t == tid ? x0 : ev_val[t];
            __syncthreads();
            for (int j = tid; j < ne; j += nt) {  // both pi(v) and lambda(v) take the evidence vector (:68-73)
                const int v = j == tid ? v0 : ev_node[j], o = j == tid ? o0 : ev_off[j];
                const int lo = a.node_off[v], hi = a.node_off[v + 1];
                for (int i = 0; i < hi - lo; ++i) {
                    L.frz[lo + i] = 1;
                    if (s == 0) {
                        const double x = vals_in_lds ? L.stg[o + i] : ev_val[o + i];
                        L.npi[c0 * a.N + lo + i] = x;
                        L.nlam[c0 * a.N + lo + i] = x;
                    }
                }
            }
            __syncthreads();
            if (s == 0 && vals_in_lds)
                for (int t = tid; t < nval; t += nt) L.stg[t] = 0.0;  // the padding of the runs is zero again
        }
        for (int x = tid; x < a.M; x += nt) {
            L.pi[c0 * a.M + x] = s == 0 ? 1.0 : state[x];
            L.lam[c0 * a.M + x] = s == 0 ? 1.0 : state[a.M + x];
        }
        if (tid < 32) L.red[tid] = 0ull;
    }
    __syncthreads();

    int done = 0;
    double r_last = 0.0;
    for (;;) {
        const int cur = s & 1;
        const double* pi_cur = L.pi + cur * a.M;
        const double* lam_cur = L.lam + cur * a.M;
        const double* npi_cur = L.npi + cur * a.N;
        const double* nlam_cur = L.nlam + cur * a.N;
        double* pi_new = L.pi + (cur ^ 1) * a.M;
        double* lam_new = L.lam + (cur ^ 1) * a.M;
        double* npi_new = L.npi + (cur ^ 1) * a.N;
        double* nlam_new = L.nlam + (cur ^ 1) * a.N;
        double wres = 0.0;
        SMALL_STAMP(0);
        // ---- phase 1: entry items
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            if (r < a.re) small_entry_any<kTermsInRegs>(e_mm[r], L, pi_cur, nlam_cur, ent[r], ecpt[r], treg[r]);
        SMALL_STAMP(1);
        __syncthreads();
        SMALL_STAMP(2);
        // ---- phase 2a: accumulator items
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (r >= a.rb) break;
            const SmallSlot q = bs[r];
            const int kind = int(q.z & 0xffu);
            const bool on = kind != 0;
            const int base = int(q.x & 0xffffu);
            const int out_idx = int(q.y & 0xffffu), k = int((q.y >> 16) & 0xffu), at = lane - int(q.y >> 24);
            // what the stores behind the sum need of the old state is requested before it
            const double old = kind == 1 ? npi_cur[out_idx] : lam_cur[out_idx];
            const bool frozen = L.frz[out_idx] != 0;
            // The run is added strictly front to back (the reference's order): the chain of dependent additions IS the
            // phase's critical path.  Runs are padded with zeros to the wave's common length (bn_small_plan.cpp), so a step
            // is four loads at immediate offsets -- issued one step ahead -- and four additions, nothing else.
            const double* ptr = L.stg + base;
            const int n4 = b_rmax[r];  // a multiple of 4
            double acc = 0.0;
            double x0 = ptr[0], x1 = ptr[1], x2 = ptr[2], x3 = ptr[3];
            int r0 = 0;
            for (; r0 + 8 <= n4; r0 += 8) {
                const double y0 = ptr[r0 + 4], y1 = ptr[r0 + 5], y2 = ptr[r0 + 6], y3 = ptr[r0 + 7];
                acc += x0; acc += x1; acc += x2; acc += x3;
                x0 = ptr[r0 + 8]; x1 = ptr[r0 + 9]; x2 = ptr[r0 + 10]; x3 = ptr[r0 + 11];
                acc += y0; acc += y1; acc += y2; acc += y3;
            }
            if (r0 < n4) { acc += x0; acc += x1; acc += x2; acc += x3; }  // an odd number of steps: the last four are loaded already
            if (r == 0) SMALL_STAMP(6);
            double* buf = kind == 1 ? npi_new : lam_new;
            const double val = small_normalize(buf, on, out_idx, k, at, acc, b_kmax[r]);
            if (kind == 1) npi_new[out_idx] = frozen ? old : val;  // evidence nodes are never updated (:177)
            if (kind == 2) {
                lam_new[out_idx] = val;
                wres = res_acc(wres, fabs(val - old));
            }
        }
        SMALL_STAMP(7);
        // ---- phase 2b: product items.  They read the old state only, so they could run on either side of the barrier: here,
        // because the plan gives the waves with the longest runs the cheapest products (bn_small_plan.cpp).
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (r >= a.rc) break;
            const SmallSlot q = cs[r];
            const int kind = int(q.z & 0xffu), skip = int((q.z >> 8) & 0xffffu);
            const bool on = kind != 0;
            const int cl = int(q.x & 0xffffu), deg = int(q.x >> 16);
            const int out_idx = int(q.y & 0xffffu), k = int((q.y >> 16) & 0xffu), at = lane - int(q.y >> 24);
            const double old = kind == 4 ? pi_cur[out_idx] : nlam_cur[out_idx];
            const bool frozen = L.frz[out_idx] != 0;
            // lambda(v): from 1.0 (:220-238); pi-message: from pi(v)[i] (:202-218); children in ascending order
            double val = kind == 4 ? npi_cur[q.w & 0xffffu] : 1.0;
            for (int x0 = 0; x0 < c_dmax[r]; x0 += 4) {
                double f[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool has = x0 + u < deg;
                    const int cb = L.clist[has ? cl + x0 + u : 0];
                    f[u] = lam_cur[has ? cb + at : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) val *= (x0 + u < deg && x0 + u != skip) ? f[u] : 1.0;  // x * 1.0 == x
            }
            double* buf = kind == 4 ? pi_new : nlam_new;
            val = small_normalize(buf, on, out_idx, k, at, val, c_kmax[r]);
            if (kind == 3) nlam_new[out_idx] = frozen ? old : val;  // evidence nodes are never updated (:177)
            if (kind == 4) {
                pi_new[out_idx] = val;
                wres = res_acc(wres, fabs(val - old));
            }
        }
        SMALL_STAMP(3);
        // maximum_difference (:105-131): the wave's maximum (bit patterns of non-negative doubles order like the values)
        // -> the wave's LDS word; after the barrier every wave reduces the same 16 words
        const unsigned long long bits = wave_umax64_dpp((unsigned long long)__double_as_longlong(wres));
        if (lane == 0) L.red[cur * 16 + wave] = bits;
        SMALL_STAMP(4);
        __syncthreads();
        const unsigned long long mx = wave_umax64_dpp<true>(L.red[cur * 16 + (lane & 15)]);
        double rr = __longlong_as_double((long long)mx);
        rr = rr < DBL_MIN ? DBL_MIN : rr;
        r_last = rr;
        SMALL_STAMP(5);
        if (tid == 0 && s < b.res_cap) b.res_hist[s] = rr;
        ++s;
        if (rr < a.eps) { done = 1; break; }                                 // strict < (:147)
        if (a.max_sweeps > 0 && s >= a.max_sweeps) { done = 2; break; }
        if (s - a.sweep_begin >= a.budget) break;                            // the host continues in another launch
    }

    // ---- the state the run stopped in -> memory; belief = normalize(pi % lambda) (:151-158)
    const int fin = s & 1;
    for (int x = tid; x < a.M; x += nt) { state[x] = L.pi[fin * a.M + x]; state[a.M + x] = L.lam[fin * a.M + x]; }
    for (int y = tid; y < a.N; y += nt) { state[2 * a.M + y] = L.npi[fin * a.N + y]; state[2 * a.M + a.N + y] = L.nlam[fin * a.N + y]; }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r >= a.rc) break;
        const SmallSlot q = cs[r];
        const bool on = (q.z & 0xffu) == 3;  // the lambda(v) items: one per node-vector element
        const int out_idx = int(q.y & 0xffffu), k = int((q.y >> 16) & 0xffu), at = lane - int(q.y >> 24);
        const double val = on ? L.npi[fin * a.N + out_idx] * L.nlam[fin * a.N + out_idx] : 0.0;
        const double bel = small_normalize(L.stg, on, out_idx, on ? k : 0, at, val, c_kmax[r]);
        if (on) b.beliefs[out_idx] = bel;
    }
    if (tid == 0) {
        Ctl* h = a.host_ctl + set;
        h->last_res = r_last; h->n_sweeps = s; h->t_first = t_first; h->t_last = wall_clock64();
        h->run_id = a.run_id; h->done = done;
    }
}

// once per device before the first launch: the kernel's dynamic LDS may exceed the 64 KiB default
int prepare_bp_small() {
    (void)hipGetLastError();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bp_small_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallLdsBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(bp_small_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(bp_small_kernel<kSmallMaxRounds>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallLdsBytes);
    return e == hipSuccess ? 0 : int(e);
}

int launch_bp_small(const SmallArgs& a, int waves, size_t lds_bytes, int n_sets, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    const dim3 grid(n_sets > 1 ? n_sets : 1), block(waves * kWave);
    const int rounds = a.re > a.rb ? (a.re > a.rc ? a.re : a.rc) : (a.rb > a.rc ? a.rb : a.rc);
    if (rounds <= 1) hipLaunchKernelGGL(bp_small_kernel<1>, grid, block, lds_bytes, (hipStream_t)stream, a);
    else if (rounds <= 2) hipLaunchKernelGGL(bp_small_kernel<2>, grid, block, lds_bytes, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(bp_small_kernel<kSmallMaxRounds>, grid, block, lds_bytes, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

}  // namespace bnmi

#ifdef BN_TILE_CLOCK
extern "C" int bn_debug_small_clock(unsigned long long* out) {
    return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(bnmi::g_small_clock), sizeof(unsigned long long) * bnmi::kSmallMaxWaves * 8));
}
#endif

// bn_mid.hip -- networks too large for ONE workgroup's LDS but small enough for a few dozen: the items of bn_small.hip
// (CPT entries, accumulator elements, product elements -- bn_small.hpp) spread over up to 128 workgroups by contiguous
// node ranges, one launch for the whole run.  Reference: belief_propagation.hpp:33-158.
//
// What differs from bn_small.hip: the STATE (pi / lambda-messages, node vectors, evidence marks) lives in device memory
// and is read and written with agent-scope accesses (sc1: through to L2 / memory, coherent across the XCDs); only the
// staged terms, the parent terms and the child lists of a workgroup's own nodes are in its LDS.  An iteration is
//   entries (gathers from memory) -> staging (LDS) -> s_barrier -> accumulators + products -> stores to memory,
//   the workgroup's maximum_difference -> one atomic max -> GRID barrier (a flag word per workgroup: it stores the
//   generation there, its first wave reads all flags) -> every workgroup reads the same word: same stop decision.
// Sums and products keep the reference's order: bit-identical to the oracle, like bn_small.hip.
// Every wait is bounded: a workgroup that gives up raises *abort (page-locked host word) and leaves; the host redoes the
// run with one launch per sweep.
#include "bn_small.hpp"
#include "bn_tiles.hpp"
#include "bn_small_dev.hpp"

namespace bnmi {

__device__ __forceinline__ double ld_state(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_state(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// diagnostic builds (make EXTRA=-DBN_TILE_CLOCK): thread 0 of every workgroup records the 100 MHz clock at the phase
// boundaries of iteration 3 (scripts/experiments/mid_clock.py prints them)
#ifdef BN_TILE_CLOCK
__device__ unsigned long long g_mid_clock[kMidMaxParts][8];
#define MID_STAMP(k) do { if (tid == 0 && blockIdx.y == 0 && s == a.sweep_begin + 3) g_mid_clock[blockIdx.x][k] = wall_clock64(); } while (0)
#else
#define MID_STAMP(k) do { } while (0)
#endif

struct MidLds {
    double* stg;        // [T] this workgroup's staged terms
    uint32_t* term;     // [TT]
    uint16_t* clist;    // [CL]
    double* scratch;    // [waves][64] normalisation: a line per wave
    unsigned long long* red;  // [2][16] + [32]: what the last grid barrier collected
    unsigned* flag;     // [1] set by the polling thread: the grid wait gave up
};

__device__ __forceinline__ MidLds mid_carve(char* base, const MidPart& pt) {
    MidLds L;
    double* d = reinterpret_cast<double*>(base);
    L.stg = d; d += pt.T;
    L.scratch = d; d += kSmallMaxWaves * kWave;
    L.red = reinterpret_cast<unsigned long long*>(d); d += 2 * 16 + 2;   // [2][16] per-wave residuals, [32]: the barrier's maximum
    L.term = reinterpret_cast<uint32_t*>(d);
    char* c = reinterpret_cast<char*>(L.term + (((pt.TT > 0 ? pt.TT : 1) + 1) & ~1));
    L.clist = reinterpret_cast<uint16_t*>(c);
    c += (size_t(pt.CL > 0 ? pt.CL : 1) * 2 + 7) & ~size_t(7);
    L.flag = reinterpret_cast<unsigned*>(c);
    return L;
}

// One CPT entry (bn_small.hip small_entry, state read from memory)
template <int MM, bool REG>
__device__ __forceinline__ void mid_entry(const MidLds& L, const double* pi_cur, const double* nlam_cur, SmallEntry h, double c,
                                          const uint32_t (&treg)[4]) {
    if (((h.y >> 24) & 1u) == 0) return;
    const int m = int((h.y >> 16) & 0xffu), tbase = int(h.y & 0xffffu);
    const double li = ld_state(nlam_cur + (h.x & 0xffffu));
    uint32_t tw[MM > 0 ? MM : 1];
    double pj[MM > 0 ? MM : 1];
#pragma unroll
    for (int j = 0; j < MM; ++j) {
        if (REG && MM <= 4) tw[j] = treg[j < 4 ? j : 0];
        else tw[j] = j < m ? L.term[tbase + j] : 0u;
    }
#pragma unroll
    for (int j = 0; j < MM; ++j) {
        const double x = ld_state(pi_cur + (tw[j] & 0xffffu));
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
__device__ __forceinline__ void mid_entry_any(int mm, const MidLds& L, const double* pi_cur, const double* nlam_cur, SmallEntry h, double c,
                                              const uint32_t (&treg)[4]) {
    switch (mm) {
        case 0: return mid_entry<0, REG>(L, pi_cur, nlam_cur, h, c, treg);
        case 1: return mid_entry<1, REG>(L, pi_cur, nlam_cur, h, c, treg);
        case 2: return mid_entry<2, REG>(L, pi_cur, nlam_cur, h, c, treg);
        case 3: return mid_entry<3, REG>(L, pi_cur, nlam_cur, h, c, treg);
        case 4: return mid_entry<4, REG>(L, pi_cur, nlam_cur, h, c, treg);
        case 5: case 6: return mid_entry<6, REG>(L, pi_cur, nlam_cur, h, c, treg);
        default: return mid_entry<8, REG>(L, pi_cur, nlam_cur, h, c, treg);
    }
}

// normalisation of the vector whose elements sit in adjacent lanes of this wave, through the wave's scratch line (:298-311)
__device__ __forceinline__ double mid_normalize(double* line, int lane, int k, int at, double val, int kmax) {
    line[lane] = val;
    lds_fence();
    const int first = lane - at;
    double sum = 0.0;
    for (int r0 = 0; r0 < kmax; r0 += 4) {
        double x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = line[first + (r0 + q < k ? r0 + q : 0)];
#pragma unroll
        for (int q = 0; q < 4; ++q) sum += r0 + q < k ? x[q] : 0.0;  // + 0.0 past the end: a sum started from +0.0 is never -0.0
    }
    lds_fence();
    return val / sum;
}

// Grid barrier number `gen` (1, 2, ...) that also carries maximum_difference (round 5; bn_resident.hip's granule form).  Once a
// workgroup's stores are out, ONE thread stores a 16-byte granule {gen | residual high half}, {gen | residual low half} into a PACKED
// table (16 bytes per workgroup; two tables, by the parity of gen: a workgroup already past this barrier writes its next granule
// elsewhere than where a slower one still reads this one); its first wave reads all granules in one round trip -- lane l those of
// workgroups l, l + 64, l + 128, l + 192 -- until every one carries `gen`, and reduces the residual: every workgroup gets the same
// maximum and takes the same stop decision.  No atomics, no residual word of its own, nothing to reset between barriers.
// (Rounds 3-4: a flag word per workgroup on a 128-byte line of its own + one atomic max per workgroup on a shared word + a read
// of that word behind the barrier: with 213 workgroups -- `mixed10k` -- every poll touched 213 lines and the atomics queued on one.)
// Bounded: returns false when the wait gave up (or another workgroup has).  *res_out: the maximum over all workgroups' `res_bits`.
typedef unsigned mid_u32x4 __attribute__((ext_vector_type(4)));
// red_slot >= 0: the workgroup's residual is the maximum of the 16 per-wave words L.red[red_slot .. red_slot + 15], written by the waves
// before the call (the block barrier that makes every wave's stores final also makes those words visible: ONE block barrier per
// iteration in front of the granule, where the residual reduction used to have one of its own -- 0.76 us of an 7.9 us iteration);
// red_slot < 0: no residual (the barrier behind the initial state).
__device__ __forceinline__ bool mid_grid_barrier(const MidArgs& a, const MidLds& L, unsigned gen, int tid, int red_slot,
                                                 unsigned long long* res_out) {
    __builtin_amdgcn_s_waitcnt(0);   // this wave's stores are acknowledged
    __syncthreads();
    if (tid < kWave) {
        unsigned long long res_bits = 0;
        if (red_slot >= 0) res_bits = wave_umax64_dpp<true>(L.red[red_slot + (tid & 15)]);
        unsigned long long* tbl = reinterpret_cast<unsigned long long*>(a.bar + 32) + (gen & 1u) * (2 * kMidMaxParts);
        // The second granule word also carries the workgroup's ARRIVAL TIME (16 bits of the 100 MHz clock above the low 16 bits of the
        // generation -- enough to tell a torn pair: a slot is reused every second generation).  The work of an iteration is static, so
        // the lag of the LAST workgroup behind this one repeats from iteration to iteration: the first poll is placed where that
        // workgroup is expected (own arrival + the previous barrier's lag + a margin) instead of at the own arrival -- a poll is a
        // ~1 us round trip, and one that leaves just before the last granule becomes visible costs a whole second trip (the form
        // bn_dag.hip and bn_resident.hip have had since round 4; DESIGN section 7 listed it as not yet applied here).
        const unsigned long long t_arr = wall_clock64();
        const unsigned own16 = unsigned(t_arr) & 0xffffu;
        if (tid == 0) {
            __hip_atomic_store(tbl + 2 * blockIdx.x, ((unsigned long long)gen << 32) | unsigned(res_bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(tbl + 2 * blockIdx.x + 1, ((unsigned long long)((own16 << 16) | (gen & 0xffffu)) << 32) | unsigned(res_bits), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.first_poll_delay >= 0 && a.nparts > 1) {
            const unsigned long long first_at = t_arr + L.red[33] + (unsigned long long)a.first_poll_delay;
            while (wall_clock64() < first_at) __builtin_amdgcn_s_sleep(1);
        }
        static_assert(kMidMaxParts <= 4 * kWave, "a poll reads every workgroup's granule: lane l those of workgroups l, l + 64, l + 128, l + 192");
        const int np = a.nparts;
        const unsigned voff = unsigned(tid) * 16u;
        const unsigned long long t0 = wall_clock64();
        unsigned polls = 0, give_up = 0;
        unsigned long long mx = 0;
        int late = 0;
        for (;;) {
            mid_u32x4 r0, r1, r2, r3;   // words of a granule: {residual high half, generation, residual low half, arrival time << 16 | generation & 0xffff}
            if (np <= kWave)
                asm volatile("global_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r0) : "v"(voff), "s"(tbl) : "memory");
            else
                asm volatile("global_load_dwordx4 %0, %4, %5 sc1\n\t"
                             "global_load_dwordx4 %1, %4, %5 offset:1024 sc1\n\t"
                             "global_load_dwordx4 %2, %4, %5 offset:2048 sc1\n\t"
                             "global_load_dwordx4 %3, %4, %5 offset:3072 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(voff), "s"(tbl) : "memory");
            bool mine = true;
            mx = 0;
            late = 0;
            auto take = [&](const mid_u32x4& r, int part) {
                if (part < np) {
                    mine = mine && r.y == gen && (r.w & 0xffffu) == (gen & 0xffffu);
                    const unsigned long long v = (unsigned long long)r.x << 32 | r.z;
                    mx = v > mx ? v : mx;
                    const int d = int(short((r.w >> 16) - own16));   // that workgroup's arrival after this one's, ticks (wraps every 655 us)
                    late = d > late ? d : late;
                }
            };
            take(r0, tid);
            if (np > kWave) { take(r1, tid + kWave); take(r2, tid + 2 * kWave); take(r3, tid + 3 * kWave); }
            if (__all(mine) != 0) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++polls & 255u) == 0) {
                const unsigned ab = __hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (wall_clock64() - t0 > a.timeout_ticks || ab != 0) {
                    if (tid == 0) __hip_atomic_store(a.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    give_up = 1;
                    break;
                }
            }
        }
        mx = wave_umax64_dpp(mx);
        const unsigned lag = wave_umax32_dpp(unsigned(late));   // (late >= 0) the latest arrival of this barrier: the next one's prediction
        if (tid == 0) { *L.flag = give_up; L.red[32] = mx; L.red[33] = give_up ? 0ull : (unsigned long long)(lag < 400u ? lag : 400u); }
    }
    __syncthreads();
    *res_out = L.red[32];
    return *L.flag == 0;
}

template <int ROUNDS>
__global__ __launch_bounds__(kSmallMaxWaves * kWave) void bp_mid_kernel(MidArgs a_in) {
    extern __shared__ __attribute__((aligned(16))) char mid_lds[];
    MidArgs a = a_in;
    BpBuffers b = a.b;
    {   // this launch's set blockIdx.y: its slice of the batch's buffers, its state slot
        const int set = a.set_base + int(blockIdx.y), slot = a.slot_base + int(blockIdx.y);
        shift_to_set(b, a.sets, set);
        a.host_ctl += set;
        if (a.ev_meta) a.ev_meta += 8 * set;
        a.pi += slot * a.state_stride; a.lam += slot * a.state_stride; a.npi += slot * a.state_stride; a.nlam += slot * a.state_stride;
        a.frz += int64_t(slot) * a.N;
        a.bar += slot * (kMidSyncBytes / 4);
        a.res += slot * (kMidSyncBytes / 8);
    }
    const MidPart pt = a.parts[blockIdx.x];
    const MidLds L = mid_carve(mid_lds, pt);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* line = L.scratch + wave * kWave;
    const unsigned long long t_first = wall_clock64();
    const bool mine = tid < pt.nt;   // (the launch is as wide as the widest part)

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
        if (mine && r < pt.re) { ent[r] = a.ent[pt.ent_off + r * pt.nt + tid]; ecpt[r] = a.ent_cpt[pt.ent_off + r * pt.nt + tid]; }
        if (mine && r < pt.rb) bs[r] = a.bslot[pt.bslot_off + r * pt.nt + tid];
        if (mine && r < pt.rc) cs[r] = a.cslot[pt.cslot_off + r * pt.nt + tid];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            treg[r][j] = 0u;
            if (kTermsInRegs && ((ent[r].y >> 24) & 1u) && j < int((ent[r].y >> 16) & 0xffu)) treg[r][j] = a.term[pt.term_off + (ent[r].y & 0xffffu) + j];
        }
        e_mm[r] = wave_imax(((ent[r].y >> 24) & 1u) ? int((ent[r].y >> 16) & 0xffu) : 0);
        b_rmax[r] = wave_imax(int(bs[r].x >> 16));
        b_kmax[r] = wave_imax(bs[r].z != 0 ? int((bs[r].y >> 16) & 0xffu) : 0);
        c_dmax[r] = wave_imax(int(cs[r].x >> 16));
        c_kmax[r] = wave_imax((cs[r].z & 0xffu) != 0 ? int((cs[r].y >> 16) & 0xffu) : 0);
    }
    for (int t = tid; t < pt.TT; t += nt) L.term[t] = a.term[pt.term_off + t];
    for (int t = tid; t < pt.CL; t += nt) L.clist[t] = a.clist[pt.clist_off + t];
    for (int t = tid; t < pt.T; t += nt) L.stg[t] = 0.0;  // the padding of the runs stays zero for the whole run
    if (tid < 34) L.red[tid] = 0ull;   // ([33]: the barrier's lag prediction)
    if (tid == 0) *L.flag = 0u;

    // ---- initial state (:33-73) of this workgroup's nodes [v0, v1): their vectors, the messages on their in-edges
    int s = a.sweep_begin;
    const int y0 = a.node_off[pt.v0], y1 = a.node_off[pt.v1];
    const int x0 = a.msg_first[pt.v0], x1 = a.msg_first[pt.v1];
    if (s == 0) {
        for (int y = y0 + tid; y < y1; y += nt) {
            bool frozen = false;
            double ev = 0.0;
            if (a.ev_mode == 0) {   // marks and vectors bp_evidence_kernel left in the tile buffers
                frozen = b.frozen[a.nv_slot[y]] == b.frozen_mark;
                ev = b.node0[a.nv_idx[y]];
            }
            a.frz[y] = frozen ? 1 : 0;
            st_state(a.npi + y, frozen ? ev : a.npi_init[y]);
            st_state(a.nlam + y, frozen ? ev : 1.0);
        }
        for (int x = x0 + tid; x < x1; x += nt) { st_state(a.pi + x, 1.0); st_state(a.lam + x, 1.0); }
        if (a.ev_mode != 0) {   // the evidence arrays themselves: every workgroup walks the list and takes its own nodes
            const int32_t* meta = a.ev_meta;
            const int ne = meta ? meta[0] : a.ev_ne;
            const int32_t* ev_node = a.ev_node + (meta ? meta[1] : 0);
            const int32_t* ev_off = a.ev_off + (meta ? meta[2] : 0);
            const double* ev_val = a.ev_val + (meta ? meta[3] : 0);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            for (int j = tid; j < ne; j += nt) {
                const int v = ev_node[j];
                if (v < pt.v0 || v >= pt.v1) continue;
                const int o = ev_off[j], lo = a.node_off[v], hi = a.node_off[v + 1];
                for (int i = 0; i < hi - lo; ++i) {   // both pi(v) and lambda(v) take the evidence vector (:68-73)
                    const double x = ev_val[o + i];
                    a.frz[lo + i] = 1;
                    st_state(a.npi + lo + i, x);
                    st_state(a.nlam + lo + i, x);
                }
            }
        }
    }
    unsigned gen = 1;
    unsigned long long all_res = 0;
    bool alive = mid_grid_barrier(a, L, gen, tid, -1, &all_res);
    // the evidence marks of this thread's pi(v) / lambda(v) items do not change during a run: read once (a byte from memory in
    // front of every store was a round trip on each phase's critical path)
    bool frz_b[ROUNDS], frz_c[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        frz_b[r] = (bs[r].z & 0xffu) == 1 && a.frz[bs[r].y & 0xffffu] != 0;
        frz_c[r] = (cs[r].z & 0xffu) == 3 && a.frz[cs[r].y & 0xffffu] != 0;
    }

    int done = 0;
    double r_last = 0.0;
    while (alive) {
        const int cur = s & 1;
        const double* pi_cur = a.pi + cur * a.M;
        const double* lam_cur = a.lam + cur * a.M;
        const double* npi_cur = a.npi + cur * a.N;
        const double* nlam_cur = a.nlam + cur * a.N;
        double* pi_new = a.pi + (cur ^ 1) * a.M;
        double* lam_new = a.lam + (cur ^ 1) * a.M;
        double* npi_new = a.npi + (cur ^ 1) * a.N;
        double* nlam_new = a.nlam + (cur ^ 1) * a.N;
        double wres = 0.0;
        // ---- everything phase 2 needs of the OLD state is requested now (its previous values for the residual, the pi(v)
        // element a pi-message starts from, the first four children's lambda-messages): a trip to L2 costs about as much as a
        // whole phase here, and these travel while the entry items run
        MID_STAMP(0);
        constexpr bool kPreload = ROUNDS == 1;   // (ROUNDS == 2 as well, 103 -> 127 VGPRs: 8.43 against 8.2-8.3 us per sweep on mixed10k -- the entry phase pays what the accumulators gain; round 6)
        double pre_b[ROUNDS], pre_c[ROUNDS], pre_v[ROUNDS], pre_f[ROUNDS][4];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            pre_b[r] = pre_c[r] = 0.0; pre_v[r] = 1.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) pre_f[r][u] = 1.0;
            if (!kPreload) continue;
            if (r < pt.rb) {
                const int kind = int(bs[r].z & 0xffu), out_idx = int(bs[r].y & 0xffffu);
                if (kind == 1) pre_b[r] = ld_state(npi_cur + out_idx);
                if (kind == 2) pre_b[r] = ld_state(lam_cur + out_idx);
            }
            if (r < pt.rc) {
                const SmallSlot q = cs[r];
                const int kind = int(q.z & 0xffu), out_idx = int(q.y & 0xffffu), at = lane - int(q.y >> 24);
                const int cl = int(q.x & 0xffffu), deg = int(q.x >> 16);
                if (kind == 4) { pre_c[r] = ld_state(pi_cur + out_idx); pre_v[r] = ld_state(npi_cur + (q.w & 0xffffu)); }
                if (kind == 3) pre_c[r] = ld_state(nlam_cur + out_idx);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (kind != 0 && u < deg) pre_f[r][u] = ld_state(lam_cur + L.clist[cl + u] + at);
            }
        }
        // ---- phase 1: entry items
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            if (r < pt.re) mid_entry_any<kTermsInRegs>(e_mm[r], L, pi_cur, nlam_cur, ent[r], ecpt[r], treg[r]);
        MID_STAMP(1);
        __syncthreads();
        MID_STAMP(2);
        // ---- phase 2a: accumulator items
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (r >= pt.rb) break;
            const SmallSlot q = bs[r];
            const int kind = int(q.z & 0xffu);
            const int base = int(q.x & 0xffffu);
            const int out_idx = int(q.y & 0xffffu), k = int((q.y >> 16) & 0xffu), at = lane - int(q.y >> 24);
            double old = pre_b[r];
            bool frozen = false;
            if (kind == 1) { if (!kPreload) old = ld_state(npi_cur + out_idx); frozen = frz_b[r]; }
            if (kind == 2 && !kPreload) old = ld_state(lam_cur + out_idx);
            const double* ptr = L.stg + base;
            const int n4 = b_rmax[r];  // a multiple of 4
            double acc = 0.0;
            double x0_ = ptr[0], x1_ = ptr[1], x2_ = ptr[2], x3_ = ptr[3];
            int r0 = 0;
            for (; r0 + 8 <= n4; r0 += 8) {
                const double y0_ = ptr[r0 + 4], y1_ = ptr[r0 + 5], y2_ = ptr[r0 + 6], y3_ = ptr[r0 + 7];
                acc += x0_; acc += x1_; acc += x2_; acc += x3_;
                x0_ = ptr[r0 + 8]; x1_ = ptr[r0 + 9]; x2_ = ptr[r0 + 10]; x3_ = ptr[r0 + 11];
                acc += y0_; acc += y1_; acc += y2_; acc += y3_;
            }
            if (r0 < n4) { acc += x0_; acc += x1_; acc += x2_; acc += x3_; }
            const double val = mid_normalize(line, lane, kind != 0 ? k : 0, kind != 0 ? at : 0, acc, b_kmax[r]);
            if (kind == 1) st_state(npi_new + out_idx, frozen ? old : val);  // evidence nodes are never updated (:177)
            if (kind == 2) {
                st_state(lam_new + out_idx, val);
                wres = res_acc(wres, fabs(val - old));
            }
        }
        MID_STAMP(3);
        // ---- phase 2b: product items (old state only)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (r >= pt.rc) break;
            const SmallSlot q = cs[r];
            const int kind = int(q.z & 0xffu), skip = int((q.z >> 8) & 0xffffu);
            const int cl = int(q.x & 0xffffu), deg = int(q.x >> 16);
            const int out_idx = int(q.y & 0xffffu), k = int((q.y >> 16) & 0xffu), at = lane - int(q.y >> 24);
            double old = pre_c[r];
            bool frozen = false;
            if (kind == 4 && !kPreload) old = ld_state(pi_cur + out_idx);
            if (kind == 3) { if (!kPreload) old = ld_state(nlam_cur + out_idx); frozen = frz_c[r]; }
            // lambda(v): from 1.0 (:220-238); pi-message: from pi(v)[i] (:202-218); children in ascending order
            double val = kind == 4 ? (kPreload ? pre_v[r] : ld_state(npi_cur + (q.w & 0xffffu))) : 1.0;
            for (int c0 = 0; c0 < c_dmax[r]; c0 += 4) {
                double f[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (kPreload && c0 == 0) { f[u] = pre_f[r][u]; continue; }
                    const bool has = c0 + u < deg;
                    const int cb = L.clist[has ? cl + c0 + u : 0];
                    f[u] = has ? ld_state(lam_cur + cb + at) : 1.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) val *= (c0 + u < deg && c0 + u != skip) ? f[u] : 1.0;  // x * 1.0 == x
            }
            val = mid_normalize(line, lane, kind != 0 ? k : 0, kind != 0 ? at : 0, val, c_kmax[r]);
            if (kind == 3) st_state(nlam_new + out_idx, frozen ? old : val);
            if (kind == 4) {
                st_state(pi_new + out_idx, val);
                wres = res_acc(wres, fabs(val - old));
            }
        }
        MID_STAMP(4);
        // maximum_difference (:105-131): wave -> workgroup (LDS) -> the workgroup's granule -> every workgroup reduces all granules
        const unsigned long long bits = wave_umax64_dpp((unsigned long long)__double_as_longlong(wres));
        if (lane == 0) L.red[cur * 16 + wave] = bits;
        MID_STAMP(5);
        ++gen;
        unsigned long long mx = 0;
        alive = mid_grid_barrier(a, L, gen, tid, cur * 16, &mx);
        MID_STAMP(6);
        if (!alive) break;
        double rr = __longlong_as_double((long long)mx);
        rr = rr < DBL_MIN ? DBL_MIN : rr;
        r_last = rr;
        if (blockIdx.x == 0 && tid == 0 && s < b.res_cap) b.res_hist[s] = rr;
        ++s;
        if (rr < a.eps) { done = 1; break; }                                 // strict < (:147)
        if (a.max_sweeps > 0 && s >= a.max_sweeps) { done = 2; break; }
        if (s - a.sweep_begin >= a.budget) break;                            // the host continues in another launch
    }
    if (!alive) done = -1;

    // ---- belief = normalize(pi % lambda) (:151-158) of this workgroup's nodes, from the state the run stopped in
    const int fin = s & 1;
    if (alive) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (r >= pt.rc) break;
            const SmallSlot q = cs[r];
            const bool on = (q.z & 0xffu) == 3;  // the lambda(v) items: one per node-vector element
            const int out_idx = int(q.y & 0xffffu), k = int((q.y >> 16) & 0xffu), at = lane - int(q.y >> 24);
            const double val = on ? ld_state(a.npi + fin * a.N + out_idx) * ld_state(a.nlam + fin * a.N + out_idx) : 0.0;
            const double bel = mid_normalize(line, lane, on ? k : 0, on ? at : 0, val, c_kmax[r]);
            if (on) b.beliefs[out_idx] = bel;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        Ctl* h = a.host_ctl;
        h->last_res = r_last; h->n_sweeps = s; h->t_first = t_first; h->t_last = wall_clock64();
        h->run_id = a.run_id; h->done = done;
    }
}

int prepare_bp_mid() {
    (void)hipGetLastError();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bp_mid_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallLdsBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(bp_mid_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(bp_mid_kernel<kSmallMaxRounds>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmallLdsBytes);
    return e == hipSuccess ? 0 : int(e);
}

int launch_bp_mid(const MidArgs& a, int waves, int rounds, size_t lds_bytes, int n_sets, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    const dim3 grid(a.nparts, n_sets > 1 ? n_sets : 1), block(waves * kWave);
    if (rounds <= 1) hipLaunchKernelGGL(bp_mid_kernel<1>, grid, block, lds_bytes, (hipStream_t)stream, a);
    else if (rounds <= 2) hipLaunchKernelGGL(bp_mid_kernel<2>, grid, block, lds_bytes, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(bp_mid_kernel<kSmallMaxRounds>, grid, block, lds_bytes, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

}  // namespace bnmi

#ifdef BN_TILE_CLOCK
extern "C" int bn_debug_mid_clock(unsigned long long* out) {
    return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(bnmi::g_mid_clock), sizeof(unsigned long long) * bnmi::kMidMaxParts * 8));
}
#endif

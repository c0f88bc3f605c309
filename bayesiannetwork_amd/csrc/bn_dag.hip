// bn_dag.hip -- networks of arity <= 4 (2 and 3 padded to 4: bn_dag_plan.cpp) with up to 5 parents per node: the whole belief-propagation run in ONE launch, every CPT
// entry resident in a register, a node's two roles on different waves (bn_dag.hpp).  Reference:
// bayesian/inference/belief_propagation.hpp:33-158.
//
// One iteration of the reference's while(true) loop (:75-148) for a wave = ONE round trip for its inputs (every load of
// the iteration is requested before the first is used), the arithmetic, 16-byte write-through stores, drain, arrival.
// Everything reads the OLD buffers and writes the NEW ones (the reference's new_* maps); the grid barrier between two
// iterations is bn_resident.hip's direct form: a block publishes a pair of 8-byte granules {generation | residual half},
// the first wave of every block collects all blocks' pairs in one round trip per poll and takes the stop decision (:147).
// Every wait is bounded; a wait that gives up raises `abort`, every block leaves and the host redoes the run with one
// launch per sweep.
#include "bn_dag.hpp"
#include "bn_tiles.hpp"
#include "bn_small_dev.hpp"

#include <type_traits>

namespace bnmi {

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

#ifdef BN_TILE_CLOCK
__device__ unsigned long long g_dag_clock[kDagMaxBlocks * kDagWaves][12];
#define DSTAMP(k, it) do { if ((it) == 5 && lane == 0) g_dag_clock[blockIdx.x * kDagWaves + wave][k] = wall_clock64(); } while (0)
// inside a tile's sweep, once its inputs have arrived (the wait is part of the diagnostic build only)
#define DSTAMP_INPUTS(a, s)                                                                          \
    do {                                                                                             \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
        if ((s) - (a).sweep_begin == 5 && (threadIdx.x & 63) == 0)                                   \
            g_dag_clock[blockIdx.x * kDagWaves + (threadIdx.x >> 6)][2] = wall_clock64();            \
    } while (0)
#define DSTAMP_AT(k, a, s)                                                                          \
    do {                                                                                             \
        if ((s) - (a).sweep_begin == 5 && (threadIdx.x & 63) == 0)                                   \
            g_dag_clock[blockIdx.x * kDagWaves + (threadIdx.x >> 6)][k] = wall_clock64();            \
    } while (0)
#else
#define DSTAMP(k, it) ((void)0)
#define DSTAMP_INPUTS(a, s) ((void)0)
#define DSTAMP_AT(k, a, s) ((void)0)
#endif

typedef unsigned dag_u32x4 __attribute__((ext_vector_type(4)));
// state records: 16-byte sc1 accesses through a buffer descriptor (aux 16): stores write through, loads bypass the CU's
// L1 -- no release / acquire fences around the barrier (cdna_hip_programming.md Guideline 16)
__device__ __forceinline__ double2_t dag_ld(__amdgpu_buffer_rsrc_t r, int64_t idx2) {
    return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(r, int(idx2) * 16, 0, 16));
}
__device__ __forceinline__ void dag_st(__amdgpu_buffer_rsrc_t r, int64_t idx2, double2_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dag_u32x4, v), r, int(idx2) * 16, 0, 16);
}
__device__ __forceinline__ void dag_ld4(__amdgpu_buffer_rsrc_t r, int64_t idx2, double (&v)[4]) {
    const double2_t a = dag_ld(r, idx2), b = dag_ld(r, idx2 + 1);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
__device__ __forceinline__ void dag_st4(__amdgpu_buffer_rsrc_t r, int64_t idx2, const double (&v)[4]) {
    double2_t a, b;
    a.x = v[0]; a.y = v[1]; b.x = v[2]; b.y = v[3];
    dag_st(r, idx2, a);
    dag_st(r, idx2 + 1, b);
}

struct DagSetShared {                    // per evidence set of the launch
    unsigned long long slot[kDagWaves];  // per-wave residual bit patterns
    int verdict;
    unsigned long long t_arrive;         // 100 MHz clock when this block published its granules
    int skew_ticks;                      // how long after this block the LAST block arrived in the previous iteration (10 ns ticks)
    int arrived;                         // waves of the block that have finished the set's sweep in hand (the last one publishes)
    int pub_it;                          // the iteration whose granules the block has published
    int poll_it;                         // several sets: the last iteration for whose barrier a wave of the block has taken the polling on
    int ver;                             // ... and the last one whose verdict is out, with the verdict: iteration << 2 | verdict (one LDS read tells both)
};
struct DagShared {
    DagSetShared set[kDagMaxSets];
    // per wave: 128 16-byte units through which the lanes of a lane group / of a node's parent items hand each other the records
    // they loaded -- one vector-memory instruction per lane and iteration instead of one per record and lane (a CU's eight waves
    // share ONE texture-address pipe at 64 bytes per cycle: with every lane requesting all of its node's records that pipe,
    // not the memory latency, set the length of the load phase -- 0.8 us for the first waves served, 2.6 us for the last)
    double2_t xch[kDagWaves][2 * kWave];
};
enum : int { kDagGoOn = 0, kDagConverged = 1, kDagCapped = 2, kDagAbort = 3 };

__device__ __forceinline__ double dag_residual_of(unsigned long long bits) {
    const double r = __longlong_as_double((long long)bits);
    return r < DBL_MIN ? DBL_MIN : r;  // maximum_difference starts at numeric_limits<double>::min() (:105)
}
__device__ __forceinline__ int dag_verdict_of(const DagArgs& a, double r, int n_done) {
    if (r < a.eps) return kDagConverged;                              // strict '<' (:147)
    if (a.max_sweeps > 0 && n_done >= a.max_sweeps) return kDagCapped;
    return kDagGoOn;
}

// ---- grid barrier (bn_resident.hip's direct form) ------------------------------------------------------------------
// A wave that has finished its sweep of a set does NOT wait for the block's other waves: it counts itself in (LDS) and goes on --
// with several sets per launch to the next set's sweep; the wave that completes the count publishes the block's granules.  (With
// a __syncthreads here every set-sweep of a batch cost the block's SLOWEST wave, 5.9 us on config 2; the waves with light tiles
// now run ahead through the other sets and wait once per set, where the verdict is needed.)
// the wave's share of maximum_difference as a wave-uniform bit pattern (non-negative doubles order like their bits)
__device__ __forceinline__ unsigned long long dag_wave_residual(double wres) {
    const unsigned long long bits = wave_umax64_dpp((unsigned long long)__double_as_longlong(wres));
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(bits)), hi = __builtin_amdgcn_readfirstlane(unsigned(bits >> 32));
    return (unsigned long long)hi << 32 | lo;
}
template <bool BATCH>
__device__ __forceinline__ void dag_arrive(const DagArgs& a, ResidentSync* sync, DagSetShared& sh, int it, int s, unsigned long long bits, int lane, int wave) {
    // bits: the wave's share of maximum_difference, dag_wave_residual(wres)
    if (lane == 0) sh.slot[wave] = bits;   // (in front of the drain: the LDS write passes while the stores are waited for)
    DSTAMP(3, it);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have reached memory
    DSTAMP(4, it);
    // (a single query has nothing to run ahead to: a plain block barrier, thread 0 publishes -- measured 0.3-0.8 us per sweep faster
    // there than counting the waves in)
    constexpr bool single = !BATCH;
    if (single) __syncthreads();
    if (lane == 0) {
        int before = kDagWaves - 1;
        if (!single) {
            before = __hip_atomic_fetch_add(&sh.arrived, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (wave != 0) {
            before = -1;
        }
        if (before == kDagWaves - 1) {   // the block's last wave (a single query: its first): every wave's stores are out, every slot is written
            if (!single) {
                __hip_atomic_store(&sh.arrived, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (next used by a wave that has seen this barrier's verdict)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            // (plain loads, requested together: as eight atomic loads they were eight LDS round trips in a row on the path to the
            // granule store -- 0.3 us per sweep)
            unsigned long long m = 0;
#pragma unroll
            for (int w = 0; w < kDagWaves; ++w) m = sh.slot[w] > m ? sh.slot[w] : m;
            if (a.n_blocks == 1) {
                sync->res[it] = m;
                sh.verdict = dag_verdict_of(a, dag_residual_of(m), s + 1);
                if (!single) __hip_atomic_store(&sh.ver, it << 2 | sh.verdict, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                // the granules of consecutive iterations alternate between two tables: a block already past this barrier must not
                // overwrite what a slower block still has to read
                // Second granule: {arrival time, 16 bits of the 100 MHz clock | low 16 bits of the generation (enough to tell a torn
                // pair: a slot is reused every second generation) | residual low half}.  The arrival times let every block predict
                // WHEN the last block will arrive in the next iteration -- the work per iteration is static -- and place its first
                // poll there instead of polling from its own arrival on: a poll is a ~1 us round trip, and one that leaves just
                // before the last granule becomes visible costs the block a whole second trip (fixed delays, config 2: 7.5 -> 6.6 us
                // per sweep).
                const unsigned gen = a.gen_base + unsigned(it) + 1u;
                unsigned long long* g = (it & 1) ? sync->blk_odd[blockIdx.x] : sync->blk[blockIdx.x];
                const unsigned long long now = wall_clock64();
                sh.t_arrive = now;
                __hip_atomic_store(g, ((unsigned long long)gen << 32) | unsigned(m >> 32), RLX_AGENT);
                __hip_atomic_store(g + 1, ((now & 0xffffull) << 48) | ((unsigned long long)(gen & 0xffffu) << 32) | unsigned(m), RLX_AGENT);
            }
            if (!single) __hip_atomic_store(&sh.pub_it, it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    DSTAMP(5, it);
}

// what a lane's (up to) four granule pairs say -- of blocks lane, lane + 64, lane + 128, lane + 192: true when all of them carry
// generation `gen`; acc = the largest residual among them, late = the latest arrival after this block's own (ticks)
__device__ __forceinline__ bool dag_take_granules(const dag_u32x4& r0, const dag_u32x4& r1, const dag_u32x4& r2, const dag_u32x4& r3, int lane, int nb,
                                                  unsigned gen, unsigned long long& acc, unsigned own16, int& late) {
    bool mine = true;
    acc = 0;
    late = 0;
    auto take = [&](const dag_u32x4& r, int blk) {  // words: {residual high half, generation, residual low half, arrival time << 16 | generation & 0xffff}
        if (blk < nb) {
            mine = mine && r.y == gen && (r.w & 0xffffu) == (gen & 0xffffu);
            const unsigned long long v = (unsigned long long)r.x << 32 | r.z;
            acc = v > acc ? v : acc;
            const int d = int(short((r.w >> 16) - own16));   // that block's arrival after this one's, ticks (wraps every 655 us)
            late = d > late ? d : late;
        }
    };
    take(r0, lane);
    if (nb > kWave) { take(r1, lane + kWave); take(r2, lane + 2 * kWave); take(r3, lane + 3 * kWave); }
    return mine;
}
// lane l requests the pairs of blocks l, l + 64, l + 128, l + 192 back to back (one round trip) -- bn_resident.hip sweep_granules
__device__ __forceinline__ bool dag_sweep_granules(const unsigned long long* tbl, int lane, int nb, unsigned gen, unsigned long long& acc,
                                                   unsigned own16, int& late) {
    static_assert(kDagMaxBlocks == 4 * kWave, "four pairs per lane cover the table");
    const unsigned voff = unsigned(lane) * 16u;
    dag_u32x4 r0, r1 = {0u, 0u, 0u, 0u}, r2 = r1, r3 = r1;
    if (nb <= kWave)
        asm volatile("global_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r0) : "v"(voff), "s"(tbl) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %4, %5 sc1\n\t"
                     "global_load_dwordx4 %1, %4, %5 offset:1024 sc1\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:2048 sc1\n\t"
                     "global_load_dwordx4 %3, %4, %5 offset:3072 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(voff), "s"(tbl) : "memory");
    return dag_take_granules(r0, r1, r2, r3, lane, nb, gen, acc, own16, late);
}
// every block has arrived at the barrier of iteration `it` of a set: block 0 files the residual, the verdict goes out to the block's waves
template <bool BATCH>
__device__ __forceinline__ void dag_publish_verdict(const DagArgs& a, ResidentSync* sync, DagSetShared& sh, int it, unsigned long long m_lane, int late,
                                                    bool ok, int lane) {
    {   // the latest arrival of this iteration, relative to this block's: the next iteration's prediction
        const int mx = int(wave_umax32_dpp(unsigned(late)));   // (late >= 0)
        if (lane == 0) sh.skew_ticks = ok ? (mx < 400 ? mx : 400) : 0;   // (bounded: 4 us)
    }
    const unsigned long long m = wave_umax64_dpp(m_lane);
    if (lane == 0) {
        if (blockIdx.x == 0 && ok) __hip_atomic_store(&sync->res[it], m, RLX_AGENT);
        sh.verdict = ok ? dag_verdict_of(a, dag_residual_of(m), a.sweep_begin + it + 1) : kDagAbort;
        if constexpr (BATCH) __hip_atomic_store(&sh.ver, it << 2 | sh.verdict, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// Several sets per launch (BATCH): no block barrier here either.  The FIRST wave of the block to need this barrier's verdict does the
// polling (an LDS fetch-max on poll_it decides who), publishes the verdict (`ver`); the others wait for it in LDS.  A wave with a
// light tile is thus never held up by the block's slower waves -- it sweeps the next sets while they finish this one -- and a set's
// sweep costs the block what its slowest wave needs for the arithmetic, not that plus a block barrier per set.
template <bool BATCH>
__device__ __forceinline__ int dag_wait(const DagArgs& a, ResidentSync* sync, DagSetShared& sh, int it) {
    bool poller = threadIdx.x < kWave;
    if constexpr (BATCH) {
        // (the block's faster waves have been here as a rule and the verdict is out: one LDS read instead of the three that the
        // claim, the wait and the verdict take -- the slowest wave's set-turn sets the pace of the whole launch)
        const int out = __hip_atomic_load(&sh.ver, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((out >> 2) == it) return out & 3;
        int before = it;
        if ((threadIdx.x & (kWave - 1)) == 0) before = __hip_atomic_fetch_max(&sh.poll_it, it, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        poller = __builtin_amdgcn_readfirstlane(before) < it;
    }
    if (a.n_blocks > 1 && poller) {
        int lane = int(threadIdx.x & (kWave - 1));
        asm volatile("" : "+v"(lane));  // (keeps the per-lane granule addresses out of the iteration loop's live registers)
        const unsigned gen = a.gen_base + unsigned(it) + 1u;
        const int nb = a.n_blocks;
        const unsigned long long* tbl = (it & 1) ? &sync->blk_odd[0][0] : &sync->blk[0][0];
        unsigned long long m = 0;
        bool ok = true;
        const unsigned long long t0 = wall_clock64();
        // (this block's own granules first: its slower waves may still be sweeping)
        while (BATCH && __hip_atomic_load(&sh.pub_it, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != it) {
            if (wall_clock64() - t0 > a.timeout_ticks) break;   // (the polls below then give up in their own way)
            __builtin_amdgcn_s_sleep(1);
        }
        // first poll: when the last block is expected to arrive (this block's arrival + the previous iteration's skew) + a margin
        const unsigned long long t_arr = sh.t_arrive;
        const unsigned own16 = unsigned(t_arr) & 0xffffu;
        const unsigned long long first_at = t_arr + (unsigned long long)(sh.skew_ticks + a.first_poll_delay);
        while (wall_clock64() < first_at) __builtin_amdgcn_s_sleep(1);
        int late = 0;
        for (unsigned n = 1;; ++n) {
            if (__all(dag_sweep_granules(tbl, lane, nb, gen, m, own16, late)) != 0) break;
            if ((n & 7u) == 0) {   // the abort word and the clock are round trips of their own: every 8th poll
                if (__hip_atomic_load(&a.sync->abort, RLX_AGENT) != 0) { ok = false; break; }
                if (wall_clock64() - t0 > a.timeout_ticks) {
                    if (__all(dag_sweep_granules(tbl, lane, nb, gen, m, own16, late)) != 0) break;
                    __hip_atomic_store(&a.sync->abort, 1u, RLX_AGENT);
                    ok = false;
                    break;
                }
            }
            for (int z = 0; z < a.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
        }
        dag_publish_verdict<BATCH>(a, sync, sh, it, m, late, ok, lane);
    }
    if constexpr (!BATCH) {
        __syncthreads();
        return sh.verdict;
    } else {
        const unsigned long long t0 = wall_clock64();
        int out;
        for (unsigned n = 1; ((out = __hip_atomic_load(&sh.ver, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 2) != it; ++n) {
            if ((n & 63u) == 0 && wall_clock64() - t0 > 2 * a.timeout_ticks) {   // (the poller gives up first and says so here)
                __hip_atomic_store(&a.sync->abort, 1u, RLX_AGENT);
                return kDagAbort;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        return out & 3;
    }
}

// The launch's loop over iterations.  phase(s): one sweep of this wave's tile(s), returns the wave's share of
// maximum_difference; finalize(n, done): the run stopped after n sweeps (done = kDagConverged / kDagCapped) or the launch's
// budget ran out (done = 0).  false: a bounded wait gave up.
// Several evidence sets (bn_bp_run_batch): the sets take turns inside an iteration -- sweep of set A, arrival at A's barrier, sweep
// of set B, arrival at B's, ... -- so the ~3.4 us a barrier needs to complete are spent on the other sets' sweeps, and ONE set of CPT
// registers serves them all.  Every set has its own state, marks, barrier words, residual history and control block, and stops
// on the sweep its single run stops on (same arithmetic: same bits).
// With several sets the sweep of a set is cut in two: request(q, s) asks for its inputs, finish(q, s) does the rest and returns the
// wave's residual.  The arrival of the PREVIOUS set-turn is published between the two: its `s_waitcnt vmcnt(0)` then covers that
// turn's stores and this turn's loads together -- a set-turn costs the wave max(store drain, load round trip) + arithmetic instead
// of their sum.  (The barrier this turn waits on belongs to an earlier turn than the one whose arrival is still held back, as long
// as another set is live; with one set left the arrival goes out before the wait.)  What was tried on top of this and dropped --
// looks ahead at the next turns' barriers and inputs fetched a turn ahead through LDS -- is in EXPERIMENTS.md R5.6.
template <bool BATCH, class Request, class Finish, class Finalize>
__device__ __forceinline__ bool dag_drive(const DagArgs& a, DagShared& sh, int lane, int wave, Request&& request, Finish&& finish, Finalize&& finalize) {
    unsigned live = BATCH ? a.set_mask : 1u;
    const int n_sets = BATCH ? a.n_sets : 1;
    int held_q = -1, held_it = 0, held_s = 0;   // several sets: the set-turn whose arrival is still to be published
    unsigned long long held_bits = 0;
    auto publish_held = [&]() {
        if (held_q >= 0) dag_arrive<BATCH>(a, a.sync + held_q, sh.set[held_q], held_it, held_s, held_bits, lane, wave);
        held_q = -1;
    };
    for (int it = 0; it <= a.budget && live != 0; ++it) {  // the pass it == budget only collects the verdicts
        const int s = a.sweep_begin + it;
        for (int q = 0; q < n_sets; ++q) {
            if (((live >> q) & 1u) == 0) continue;
            ResidentSync* sync = a.sync + q;
            DagSetShared& ss = sh.set[q];
            int v = kDagGoOn;
            DSTAMP(0, it);
            if (it > 0) {
                if constexpr (BATCH) {
                    if (held_q == q) publish_held();   // the only set left: its own arrival is what this wait waits for
                }
                v = dag_wait<BATCH>(a, sync, ss, it - 1);
                if (v == kDagAbort) return false;
            }
            DSTAMP(1, it);
            if (v != kDagGoOn || it == a.budget) {
                if constexpr (BATCH) publish_held();
                const int done = v != kDagGoOn ? v : 0;
                finalize(q, s, done);
                if (blockIdx.x == 0 && wave == 0) {  // report: residual history, outcome
                    int q0 = lane;
                    asm volatile("" : "+v"(q0));
                    double* hist = a.b.res_hist + int64_t(q) * a.res_hist_stride;
                    for (int x = q0; x < it; x += kWave)
                        if (a.sweep_begin + x < a.b.res_cap)
                            hist[a.sweep_begin + x] = dag_residual_of(__hip_atomic_load(&sync->res[x], RLX_AGENT));
                    if (lane == 0) {
                        Ctl* hc = a.host_ctl + q;
                        hc->last_res = it > 0 ? dag_residual_of(__hip_atomic_load(&sync->res[it - 1], RLX_AGENT)) : 0.0;
                        hc->n_sweeps = s;
                        hc->run_id = a.run_id;
                        hc->done = done;
                    }
                }
                live &= ~(1u << q);
                continue;
            }
            request(q, s);
            DSTAMP(9, it);
            if constexpr (BATCH) publish_held();
            const unsigned long long bits = dag_wave_residual(finish(q, s));
            if constexpr (BATCH) {
                held_q = q; held_it = it; held_s = s; held_bits = bits;
            } else {
                dag_arrive<BATCH>(a, sync, ss, it, s, bits, lane, wave);
            }
            DSTAMP(6, it);
        }
    }
    if constexpr (BATCH) publish_held();
    return true;
}

// ---- the dataflow form of a single query: no grid barrier (bn_dag.hpp, DagFlowSync; the protocol of bn_resident.hip's flow form) ----
// Why here: a sweep of config 2 under the barrier is the slowest wave of the whole grid (a 4-parent child tile: round trip, 1.8 us of
// arithmetic, drain) PLUS the barrier's hand-off (every block's granules gathered by every block: 2.5 us from a wave's arrival to
// the verdict) -- 5.6 us, 0.87 of the wave cycles waiting.  Jacobi iteration i + 1 of a tile reads what its NEIGHBOUR tiles wrote in
// iteration i and overwrites what they read in i, nothing else (belief_propagation.hpp:78-101 read the old maps, :135-143 commit):
// it may start as soon as those tiles have finished i.  A heavy tile then holds up its handful of neighbours, not 1 800 waves, and
// the hand-off is one granule read by the lanes of one wave.  The stop decision (:147) needs the maximum over ALL tiles, so it
// lags: the service block gathers the granules of i while the tiles compute i + 1; a tile starts i + 2 only once the verdict of i is
// out.  When it says "stop after i" a tile has at most executed i + 1 as well -- into the OTHER buffer: the state the run ends in is
// intact, and the beliefs are formed from it.  One speculative iteration per run is the price.
__device__ __forceinline__ unsigned long long* dag_flow_slot(DagFlowSync* f, int n_tiles, int par, int tile) {
    return reinterpret_cast<unsigned long long*>(f + 1) + (size_t(par) * n_tiles + tile) * 2;
}
__device__ __forceinline__ unsigned long long dag_granule(unsigned gen, unsigned half) { return (unsigned long long)gen << 32 | half; }
__device__ __forceinline__ void dag_flow_raise_abort(const DagArgs& a) {
    const unsigned long long w = (unsigned long long)(a.gen_base + 1u) | ((unsigned long long)kDagFlowAbort << 32);
    for (int q = 0; q < 8; ++q) __hip_atomic_store(&a.flow->verdict[q].word, w, RLX_AGENT);
    if (a.host_abort) __hip_atomic_store(a.host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Waits until this tile may run iteration `it`; returns the verdict that ends the run for it (kDagFlowGoOn: run the iteration) and,
// with a stop verdict, the number of iterations of this launch the run consists of in n_it.
__device__ __forceinline__ unsigned dag_flow_wait(const DagArgs& a, int nbr, int it, int lane, int& n_it) {
    n_it = 0;
    if (it == 0) return kDagFlowGoOn;
    DagFlowSync* f = a.flow;
    const unsigned long long* vword = &f->verdict[blockIdx.x & 7].word;
    // a neighbour's slot of this parity carries EXACTLY the generation of iteration it - 1 once it has finished it: it cannot finish
    // it + 1 before this tile has finished `it`
    const unsigned want_nb = a.gen_base + unsigned(it);
    const unsigned want_v = a.gen_base + unsigned(it) - 1u;   // iteration it - 2 is decided
    const unsigned long long* g = dag_flow_slot(f, a.n_tiles, (it - 1) & 1, nbr >= 0 ? nbr : 0);
    unsigned long long t0 = 0;
    for (unsigned polls = 0;; ++polls) {
        // the neighbours' granules and the verdict word in ONE round trip
        const unsigned long long gw = nbr >= 0 ? __hip_atomic_load(g, RLX_AGENT) : 0ull;
        const unsigned long long w = __hip_atomic_load(vword, RLX_AGENT);
        const bool nb_ok = nbr < 0 || unsigned(gw >> 32) == want_nb;
        const unsigned gen = unsigned(w), kind = unsigned(w >> 32);
        const bool ours = gen - (a.gen_base + 1u) < unsigned(kDagBudget);   // published by THIS launch (generations count on across launches)
        if (ours && kind != kDagFlowGoOn) {
            n_it = int(gen - a.gen_base);
            return kind;
        }
        const bool v_ok = it < 2 || (ours && gen >= want_v);
        if (it < a.budget && v_ok && __all(nb_ok)) return kDagFlowGoOn;
        if ((polls & 31u) == 31u) {   // (the 100 MHz clock is a memory read of its own: it bounds the wait, looked at on every 32nd poll)
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > a.timeout_ticks) {
                if (lane == 0) dag_flow_raise_abort(a);
                return kDagFlowAbort;
            }
        }
        for (int z = 0; z < a.flow_sleep; ++z) __builtin_amdgcn_s_sleep(8);
    }
}
// a tile's loop over iterations.  finalize(q = 0, n, done): as in dag_drive.
template <class Request, class Finish, class Finalize>
__device__ __forceinline__ bool dag_flow_drive(const DagArgs& a, int tile, int lane, int wave, Request&& request, Finish&& finish, Finalize&& finalize) {
    const int nbr = a.nbr[int64_t(tile) * kWave + lane];
    for (int it = 0;; ++it) {
        DSTAMP(0, it);
        int n_it;
        const unsigned v = dag_flow_wait(a, nbr, it, lane, n_it);
        DSTAMP(1, it);
        if (v == kDagFlowAbort) return false;
        if (v != kDagFlowGoOn) {   // the run consists of n_it iterations of this launch (this wave has executed n_it or n_it + 1)
            finalize(0, a.sweep_begin + n_it, v == kDagFlowBudget ? 0 : int(v));
            return true;
        }
        const int s = a.sweep_begin + it;
        request(0, s);
        DSTAMP(9, it);
        const unsigned long long bits = dag_wave_residual(finish(0, s));
        DSTAMP(3, it);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have reached memory
        DSTAMP(4, it);
        if (lane == 0) {
            const unsigned gen = a.gen_base + unsigned(it) + 1u;
            unsigned long long* g = dag_flow_slot(a.flow, a.n_tiles, it & 1, tile);
            __hip_atomic_store(g, dag_granule(gen, unsigned(bits >> 32)), RLX_AGENT);
            __hip_atomic_store(g + 1, dag_granule(gen, unsigned(bits)), RLX_AGENT);
        }
        DSTAMP(6, it);
    }
}
// The service block (block n_blocks of the launch), all its waves: thread x sweeps the granule pairs of tiles x, x + 512, ... of the
// iteration in hand until they carry its generation; the block reduces, thread 0 decides and publishes.
__device__ __forceinline__ void dag_flow_service(const DagArgs& a, DagShared& sh, int lane, int wave) {
    DagFlowSync* f = a.flow;
    const int nt = a.n_tiles;
    const unsigned long long t_first = wall_clock64();
    DagSetShared& ss = sh.set[0];
    if (threadIdx.x == 0) ss.arrived = 0;   // (here: some wave's sweep gave up)
    __syncthreads();
    int n_it = 0;
    unsigned v = kDagFlowGoOn;
    for (int it = 0; it < a.budget; ++it) {
        const unsigned gen = a.gen_base + unsigned(it) + 1u;
        unsigned long long m = 0, t0 = 0;
        bool ok = true;
        for (unsigned polls = 0;; ++polls) {
            bool mine = true;
            unsigned long long acc = 0;
            for (int t = threadIdx.x; t < nt; t += kDagWaves * kWave) {
                const unsigned long long* g = dag_flow_slot(f, nt, it & 1, t);
                const unsigned long long hi = __hip_atomic_load(g, RLX_AGENT);
                const unsigned long long lo = __hip_atomic_load(g + 1, RLX_AGENT);
                mine = mine && unsigned(hi >> 32) == gen && unsigned(lo >> 32) == gen;
                const unsigned long long x = (hi << 32) | (lo & 0xffffffffull);
                acc = x > acc ? x : acc;
            }
            m = acc;
            if (__all(mine)) break;
            const unsigned long long w = __hip_atomic_load(&f->verdict[0].word, RLX_AGENT);
            if (unsigned(w >> 32) == kDagFlowAbort && unsigned(w) - (a.gen_base + 1u) < unsigned(kDagBudget)) { ok = false; break; }
            if ((polls & 31u) == 31u) {
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > a.timeout_ticks) {
                    if (lane == 0) dag_flow_raise_abort(a);
                    ok = false;
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(1);
        }
        m = wave_umax64_dpp(m);
        if (lane == 0) {
            ss.slot[wave] = m;
            if (!ok) ss.arrived = 1;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned decision = kDagFlowAbort;
            if (ss.arrived == 0) {
                unsigned long long mm = 0;
                for (int w = 0; w < kDagWaves; ++w) mm = ss.slot[w] > mm ? ss.slot[w] : mm;
                decision = unsigned(dag_verdict_of(a, dag_residual_of(mm), a.sweep_begin + it + 1));
                if (decision == kDagFlowGoOn && it == a.budget - 1) decision = kDagFlowBudget;
                __hip_atomic_store(&f->res[it], mm, RLX_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the residual is recorded before anyone learns the verdict
                const unsigned long long word = (unsigned long long)gen | ((unsigned long long)decision << 32);
                for (int q = 0; q < 8; ++q) __hip_atomic_store(&f->verdict[q].word, word, RLX_AGENT);
            }
            ss.verdict = int(decision);
        }
        __syncthreads();
        v = unsigned(ss.verdict);
        n_it = it + 1;
        if (v != kDagFlowGoOn) break;
    }
    if (wave != 0) return;
    if (v != kDagFlowAbort) {   // report: residual history, outcome, device clock
        double* hist = a.b.res_hist;
        for (int q = lane; q < n_it; q += kWave)
            if (a.sweep_begin + q < a.b.res_cap) hist[a.sweep_begin + q] = dag_residual_of(__hip_atomic_load(&f->res[q], RLX_AGENT));
    }
    if (lane == 0) {
        Ctl* hc = a.host_ctl;
        hc->last_res = (v != kDagFlowAbort && n_it > 0) ? dag_residual_of(__hip_atomic_load(&f->res[n_it - 1], RLX_AGENT)) : 0.0;
        hc->n_sweeps = a.sweep_begin + n_it;
        hc->t_first = t_first;
        hc->t_last = wall_clock64();
        hc->run_id = a.run_id;
        hc->done = v == kDagFlowAbort ? -1 : (v == kDagFlowBudget ? 0 : int(v));
    }
}

// belief = normalize(pi % lambda) (:151-158) of `node` from the state after n sweeps
__device__ __forceinline__ void dag_belief(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int node, int snode, int n, double* beliefs) {
    double pv[4], lv[4], bel[4];
    dag_ld4(rs, dag_off_npi(a.E, a.n, n & 1, snode), pv);
    dag_ld4(rs, dag_off_nlam(a.E, a.n, n & 1, snode), lv);
    double sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { bel[i] = pv[i] * lv[i]; sum += bel[i]; }   // (padding entries are 0: they add nothing)
    if (a.node_k == nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) beliefs[int64_t(node) * 4 + i] = bel[i] / sum;
    } else {
        const int kv = a.node_k[node];
        double* out = beliefs + a.node_off[node];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < kv) out[i] = bel[i] / sum;
    }
}

// bit q: `node` carries set q's evidence mark (a single query: bit 0)
__device__ __forceinline__ unsigned dag_frozen_bits(const DagArgs& a, bool active, int node) {
    // all sets' marks requested together and looked at when first used (as a loop over a.n_sets every load was waited for on the
    // spot: a memory round trip per set in front of the tile's CPT loads); a set beyond the last reads the last set's mark again
    unsigned char v[kDagMaxSets];
#pragma unroll
    for (int q = 0; q < kDagMaxSets; ++q) v[q] = a.frz[int64_t(q < a.n_sets ? q : a.n_sets - 1) * a.frz_stride + node];
    unsigned bits = 0;
#pragma unroll
    for (int q = 0; q < kDagMaxSets; ++q) bits |= (active && q < a.n_sets && v[q] == a.frz_mark ? 1u : 0u) << q;
    return bits;
}

// ---- child tile, at most two parents: one lane per node, the reference's operation order (bn_tiles.hpp tile_uniform) ----
template <int M>
struct DagChildU {
    static constexpr int K = 4, C = ipow(K, M), S = K * C, CB = (M > 0) ? C / K : 0;
    int node, ebase;          // node id; state record of its first in-edge's messages (tile-major numbering, bn_dag.hpp)
    int sdelta, estride;      // wave-uniform: slot of the node's vectors = ebase + sdelta; records between its consecutive in-edges' messages
    bool active, frozen;
    unsigned frozen_bits;   // bit q: the node is an evidence node of set q (frozen: of the set whose turn it is)
    __device__ __forceinline__ void turn(int q) { frozen = ((frozen_bits >> q) & 1u) != 0; }
    double cpt[S];  // entry cond * 4 + i
    __device__ __forceinline__ void init(const DagArgs& a, const DagTile& t, int lane) {
        const DagChildLane cl = a.cnode[t.lane_base + lane];
        active = cl.node >= 0;
        node = active ? cl.node : 0;  // idle lanes shadow node 0 and store nothing
        ebase = __builtin_amdgcn_readfirstlane(t.rec_base) + (active ? lane : 0);   // (idle lanes shadow the tile's first node)
        sdelta = __builtin_amdgcn_readfirstlane(t.slot_base - t.rec_base);
        estride = __builtin_amdgcn_readfirstlane(t.n_active);
        frozen_bits = dag_frozen_bits(a, active, node); frozen = (frozen_bits & 1u) != 0;
        const double2_t* cp = reinterpret_cast<const double2_t*>(a.cpt_img) + t.cpt_base + lane;
#pragma unroll
        for (int q = 0; q < S / 2; ++q) {
            const double2_t x = cp[q * kWave];
            cpt[2 * q] = x.x; cpt[2 * q + 1] = x.y;
        }
    }
    // the inputs of the sweep in hand: requested by request(), consumed by finish() -- with several evidence sets per launch the
    // previous set's stores drain while these loads are under way (dag_drive)
    double pim[M > 0 ? M : 1][K], lold[M > 0 ? M : 1][K], lav[K], pold[K];
    __device__ __forceinline__ void request(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int s) {
        const bool first = s == 0 && a.state_init == 0;   // (a padded network's initial state stands in memory)
        const int cur = s & 1;
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i < K; ++i) { pim[j][i] = 1.0; lold[j][i] = 1.0; }
#pragma unroll
        for (int i = 0; i < K; ++i) { lav[i] = 1.0; pold[i] = 1.0; }
        if (!first) {
#pragma unroll
            for (int j = 0; j < M; ++j) {
                dag_ld4(rs, dag_off_pim(a.E, a.n, cur, ebase + j * estride), pim[j]);
                dag_ld4(rs, dag_off_lam(a.E, a.n, cur, ebase + j * estride), lold[j]);
            }
        }
        if (!first || frozen) {  // evidence nodes hold their vector as pi and lambda in both buffers (:68-73)
            dag_ld4(rs, dag_off_nlam(a.E, a.n, cur, ebase + sdelta), lav);
            dag_ld4(rs, dag_off_npi(a.E, a.n, cur, ebase + sdelta), pold);
        }
    }
    __device__ __forceinline__ double sweep(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int s, double2_t* xch) {
        request(a, rs, s);
        return finish(a, rs, s, xch);
    }
    __device__ __forceinline__ double finish(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int s, double2_t*) {
        const int nxt = (s & 1) ^ 1;
        DSTAMP_INPUTS(a, s);
        // calculate_pi (:174-200): assignments ascending, cpt * pi-messages in ascending parent order;
        // calculate_lambda_k (:240-266): bucket out[jt][ct] receives, own state outer and assignment inner,
        // (lambda[i] * cpt) * the OTHER parents' pi-messages
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
#pragma unroll
                for (int rr = 0; rr < CB; ++rr) {
#pragma unroll
                    for (int x = 0; x < K; ++x) {
                        const int cond = rr * K + x;
                        double value = cpt[cond * K + ib];
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
                            double value = lav[ib] * cpt[cond * K + ib];
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
        normalize_k<K>(pin);
#pragma unroll
        for (int jt = 0; jt < M; ++jt) normalize_k<K>(out[jt]);
        double wres = 0.0;
#pragma unroll
        for (int jt = 0; jt < M; ++jt)
#pragma unroll
            for (int i = 0; i < K; ++i) wres = res_acc(wres, fabs(out[jt][i] - lold[jt][i]));
        if (active) {
            if (frozen) dag_st4(rs, dag_off_npi(a.E, a.n, nxt, ebase + sdelta), pold);   // evidence nodes keep pi (:177)
            else dag_st4(rs, dag_off_npi(a.E, a.n, nxt, ebase + sdelta), pin);
#pragma unroll
            for (int jt = 0; jt < M; ++jt) dag_st4(rs, dag_off_lam(a.E, a.n, nxt, ebase + jt * estride), out[jt]);
        }
        return active ? wres : 0.0;
    }
    __device__ __forceinline__ void belief(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int n, double* beliefs) {
        if (active) dag_belief(a, rs, node, ebase + sdelta, n, beliefs);
    }
};

// sum over the aligned group of G = 4, 16 or 64 lanes, left in every lane of the group: the pairing of an xor butterfly
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int G>
__device__ __forceinline__ double group_allreduce(double x) {
    x += dpp_mov_d<0xB1>(x);                       // quad_perm [1, 0, 3, 2]: lane ^ 1
    x += dpp_mov_d<0x4E>(x);                       // quad_perm [2, 3, 0, 1]: lane ^ 2
    if constexpr (G >= 16) {
        x += dpp_mov_d<0x141>(x);                  // row_half_mirror: the other quad of the half row
        x += dpp_mov_d<0x140>(x);                  // row_mirror: the other half row
    }
    if constexpr (G >= 64) {
        x += shfl_xor_d(x, 16);
        x += shfl_xor_d(x, 32);
    }
    return x;
}

// ---- child tile, M = D + 2 parents: G = 4^D lanes per node, lane g holds the 64 entries of one assignment of the D leading
// parents.  The sums over the two trailing parents (states c, d) are formed ONCE per lane and shared by the five outputs:
//   Q[c][d] = sum_i lambda(v)[i] P[c][d][i]       R[c][i] = sum_d P[c][d][i] piD[d]       S[i] = sum_c piC[c] R[c][i]
//   LC[c]   = sum_d Q[c][d] piD[d]                LD[d]   = sum_c piC[c] Q[c][d]          L    = sum_c piC[c] LC[c]
//   pi(v)[i] += w S[i],  lambda->C[c] += w LC[c],  lambda->D[d] += w LD[d]   (w = product of the lane's leading pi-message entries)
//   lambda->leading parent jt, bucket = the lane's digit jt:  += L * product of the OTHER leading entries
// 350 fp64 operations per lane where the reference's term-by-term products (bn_tiles.hpp tile_group) take 1 280; the results
// agree with the reference to rounding (its own >= 3-parent products are unordered, :253).  The group's partial sums are
// combined with shuffles exactly as in tile_group; lanes 0 .. M of a group finish one vector each.
template <int D>
struct DagChildG {
    static constexpr int K = 4, M = D + 2, G = 1 << (2 * D), NPT = kWave / G;
    static_assert(G >= M + 1, "a group has a lane per finished vector");
    int node, ebase, nl, g;   // (ebase: state record of the first in-edge's messages, tile-major numbering, bn_dag.hpp)
    int sdelta, estride;      // wave-uniform: slot of the node's vectors = ebase + sdelta; records between its consecutive in-edges' messages
    bool active, frozen;
    unsigned frozen_bits;   // bit q: the node is an evidence node of set q (frozen: of the set whose turn it is)
    __device__ __forceinline__ void turn(int q) { frozen = ((frozen_bits >> q) & 1u) != 0; }
    double cpt[64];  // entry (c * 4 + d) * 4 + i
    __device__ __forceinline__ void init(const DagArgs& a, const DagTile& t, int lane) {
        const DagChildLane cl = a.cnode[t.lane_base + lane];
        nl = lane / G; g = lane % G;
        active = cl.node >= 0;
        node = active ? cl.node : 0;
        ebase = __builtin_amdgcn_readfirstlane(t.rec_base) + (active ? nl : 0);
        sdelta = __builtin_amdgcn_readfirstlane(t.slot_base - t.rec_base);
        estride = __builtin_amdgcn_readfirstlane(t.n_active);
        frozen_bits = dag_frozen_bits(a, active, node); frozen = (frozen_bits & 1u) != 0;
        const double2_t* cp = reinterpret_cast<const double2_t*>(a.cpt_img) + t.cpt_base + lane;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const double2_t x = cp[q * kWave];
            cpt[2 * q] = x.x; cpt[2 * q + 1] = x.y;
        }
    }
    // The node's inputs -- its M pi-messages and lambda(v), 2 (M + 1) 16-byte halves -- are requested ONCE per group: lane g
    // loads half g (and g + G where the group has fewer lanes than halves), the group shares them through the wave's scratch.
    static constexpr int HN = 2 * (M + 1), LPL = (HN + G - 1) / G;
    static_assert(NPT * HN <= 2 * kWave, "the groups' halves fit the wave's scratch");
    // the inputs of the sweep in hand: requested by request(), consumed by finish() (see DagChildU)
    double fold[K];
    double2_t mine[LPL];
    __device__ __forceinline__ void request(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int s) {
        const bool first = s == 0 && a.state_init == 0;   // (a padded network's initial state stands in memory)
        const int cur = s & 1;
#pragma unroll
        for (int i = 0; i < K; ++i) fold[i] = 1.0;
#pragma unroll
        for (int q = 0; q < LPL; ++q) {
            const int h = g + q * G;
            mine[q].x = 1.0; mine[q].y = 1.0;
            if (h < 2 * M) {
                if (!first) mine[q] = dag_ld(rs, dag_off_pim(a.E, a.n, cur, ebase + (h >> 1) * estride) + (h & 1));
            } else if (h < HN) {
                if (!first || frozen) mine[q] = dag_ld(rs, dag_off_nlam(a.E, a.n, cur, ebase + sdelta) + (h - 2 * M));
            }
        }
        // lane 0 of the group finishes pi(v), lane 1 + jt the lambda-message to parent jt: each requests the previous value
        // of ITS vector now (residual :105-131; an evidence node's pi is carried over, :177)
        const bool fin_msg = g >= 1 && g <= M;
        if (fin_msg && !first) dag_ld4(rs, dag_off_lam(a.E, a.n, cur, ebase + (g - 1) * estride), fold);
        if (g == 0 && frozen) dag_ld4(rs, dag_off_npi(a.E, a.n, cur, ebase + sdelta), fold);
    }
    __device__ __forceinline__ double sweep(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int s, double2_t* xch) {
        request(a, rs, s);
        return finish(a, rs, s, xch);
    }
    __device__ __forceinline__ double finish(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int s, double2_t* xch) {
        const int nxt = (s & 1) ^ 1;
        double2_t* xn = xch + nl * HN;
#pragma unroll
        for (int q = 0; q < LPL; ++q)
            if (g + q * G < HN) xn[g + q * G] = mine[q];
        lds_fence();
        double pC[K], pD[K], lav[K];
        {
            const double2_t c0 = xn[2 * D], c1 = xn[2 * D + 1], d0 = xn[2 * D + 2], d1 = xn[2 * D + 3], l0 = xn[2 * M], l1 = xn[2 * M + 1];
            pC[0] = c0.x; pC[1] = c0.y; pC[2] = c1.x; pC[3] = c1.y;
            pD[0] = d0.x; pD[1] = d0.y; pD[2] = d1.x; pD[3] = d1.y;
            lav[0] = l0.x; lav[1] = l0.y; lav[2] = l1.x; lav[3] = l1.y;
        }
        double pfix[D > 0 ? D : 1];
#pragma unroll
        for (int j = 0; j < D; ++j)   // the entry of leading parent j's pi-message that the lane's digit j selects
            pfix[j] = reinterpret_cast<const double*>(xn)[4 * j + ((g >> (2 * (D - 1 - j))) & 3)];
        lds_fence();
        DSTAMP_INPUTS(a, s);
        double S[K], LC[K], LD[K];
#pragma unroll
        for (int i = 0; i < K; ++i) { S[i] = 0.0; LD[i] = 0.0; }
#pragma unroll
        for (int c = 0; c < K; ++c) {   // (fused multiply-adds: this form is its own rounding anyway, see the struct's comment)
            double R[K], lc = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) R[i] = 0.0;
#pragma unroll
            for (int d = 0; d < K; ++d) {
                const double* e = &cpt[(c * K + d) * K];
                const double q = __builtin_fma(lav[3], e[3], __builtin_fma(lav[2], e[2], __builtin_fma(lav[1], e[1], lav[0] * e[0])));
#pragma unroll
                for (int i = 0; i < K; ++i) R[i] = __builtin_fma(e[i], pD[d], R[i]);
                lc = __builtin_fma(q, pD[d], lc);
                LD[d] = __builtin_fma(pC[c], q, LD[d]);
            }
#pragma unroll
            for (int i = 0; i < K; ++i) S[i] = __builtin_fma(pC[c], R[i], S[i]);
            LC[c] = lc;
        }
        double L = 0.0;
#pragma unroll
        for (int c = 0; c < K; ++c) L = __builtin_fma(pC[c], LC[c], L);
        double w = 1.0;
#pragma unroll
        for (int j = 0; j < D; ++j) w *= pfix[j];
        double pp[K], ol[2][K], sf[D > 0 ? D : 1];
#pragma unroll
        for (int i = 0; i < K; ++i) { pp[i] = w * S[i]; ol[0][i] = w * LC[i]; ol[1][i] = w * LD[i]; }
#pragma unroll
        for (int jt = 0; jt < D; ++jt) {
            double x = L;
#pragma unroll
            for (int j = 0; j < D; ++j)
                if (j != jt) x *= pfix[j];
            sf[jt] = x;
        }
        DSTAMP_AT(7, a, s);
        // ---- combine inside the G-lane group
        // (xor butterflies; inside a 16-lane row as DPP moves -- quad_perm for 1 and 2, row_half_mirror and row_mirror for 4 and
        // 8: after each step the lanes of a sub-group hold the same value, so the mirrored partner carries what the xor partner
        // would: the same bits as __shfl_xor, without the LDS round trips)
#pragma unroll
        for (int i = 0; i < K; ++i) pp[i] = group_allreduce<G>(pp[i]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ct = 0; ct < K; ++ct) ol[t][ct] = group_allreduce<G>(ol[t][ct]);
        // leading parent jt: sum over the lanes that share digit jt, then collect the four buckets from the lanes whose
        // other digits are zero
        double of[D > 0 ? D : 1][K];
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
        DSTAMP_AT(8, a, s);
        // ---- lanes 0 .. M of the group finish the node: normalise (:298-311), residual, stores
        double o[K];
#pragma unroll
        for (int ct = 0; ct < K; ++ct) {
            o[ct] = pp[ct];
#pragma unroll
            for (int jt = 0; jt < M; ++jt)
                if (g == jt + 1) o[ct] = jt < D ? of[jt < D ? jt : 0][ct] : ol[jt >= D ? jt - D : 0][ct];
        }
        double wres = 0.0;
        if (active && g <= M) {
            normalize_k<K>(o);
            if (g == 0) {
                if (frozen) dag_st4(rs, dag_off_npi(a.E, a.n, nxt, ebase + sdelta), fold);
                else dag_st4(rs, dag_off_npi(a.E, a.n, nxt, ebase + sdelta), o);
            } else {
#pragma unroll
                for (int i = 0; i < K; ++i) wres = res_acc(wres, fabs(o[i] - fold[i]));
                dag_st4(rs, dag_off_lam(a.E, a.n, nxt, ebase + (g - 1) * estride), o);
            }
        }
        return wres;
    }
    __device__ __forceinline__ void belief(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int n, double* beliefs) {
        if (active && g == 0) dag_belief(a, rs, node, ebase + sdelta, n, beliefs);
    }
};

// ---- parent items (kDagParentWide: nodes with more children than a wave has lanes): a lane per pi-message (:202-218) and per lambda(v) (:220-238); the product runs over the node's children
// in ascending order, the target left out -- the reference's multiplication sequence
struct DagParent {
    static constexpr int K = 4, RC = kDagRegChildren;
    int node, snode, tedge, obeg, deg, tpos, dmax, kv;   // (tedge, the out-edges: state records; snode: slot of the node's vectors)
    bool active, frozen;
    unsigned frozen_bits;   // bit q: the node is an evidence node of set q (frozen: of the set whose turn it is)
    __device__ __forceinline__ void turn(int q) { frozen = ((frozen_bits >> q) & 1u) != 0; }
    int oe[RC];  // the first out-edges' CSR ids
    __device__ __forceinline__ void init(const DagArgs& a, const DagTile& t, int lane) {
        const DagParentLaneDev it = a.pitem[t.lane_base + lane];
        active = it.node >= 0;
        node = active ? it.node : 0;
        snode = active ? it.snode : 0;
        tedge = it.tedge;
        obeg = it.obeg;
        deg = active ? (it.deg_tpos & 0xffff) : 0;
        tpos = it.deg_tpos >> 16;  // (0xffff for the lambda(v) item: no child is left out)
        dmax = t.dmax;
        kv = a.node_k ? a.node_k[node] : 4;
        frozen_bits = dag_frozen_bits(a, active, node); frozen = (frozen_bits & 1u) != 0;
#pragma unroll
        for (int x = 0; x < RC; ++x) oe[x] = (x < deg) ? a.oedge[obeg + x] : 0;
    }
    // the inputs of the sweep in hand: requested by request(), consumed by finish() (see DagChildU)
    double acc[K], old[K], lk[RC][K];
    __device__ __forceinline__ void request(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int s) {
        const bool first = s == 0 && a.state_init == 0;   // (a padded network's initial state stands in memory)
        const int cur = s & 1;
        const bool is_msg = tedge >= 0;
#pragma unroll
        for (int i = 0; i < K; ++i) { acc[i] = i < kv ? 1.0 : 0.0; old[i] = 1.0; }   // (the empty product over a leaf's children: ones over the node's OWN states)
#pragma unroll
        for (int x = 0; x < RC; ++x)
#pragma unroll
            for (int i = 0; i < K; ++i) lk[x][i] = 1.0;
        // a pi-message starts from pi(v) (:207): the previous sweep's, or in sweep 0 the initial one (1.0, a root's CPT row, the evidence)
        if (is_msg) {
            if (!first || frozen) dag_ld4(rs, dag_off_npi(a.E, a.n, cur, snode), acc);
            else {
#pragma unroll
                for (int i = 0; i < K; ++i) acc[i] = a.npi_init[int64_t(node) * 4 + i];
            }
            if (!first) dag_ld4(rs, dag_off_pim(a.E, a.n, cur, tedge), old);
        } else if (frozen) {
            dag_ld4(rs, dag_off_nlam(a.E, a.n, cur, snode), old);  // an evidence node's lambda is carried over (:223)
        }
        if (!first) {
#pragma unroll
            for (int x = 0; x < RC; ++x)
                if (x < dmax) {  // wave-uniform
                    if (x < deg) dag_ld4(rs, dag_off_lam(a.E, a.n, cur, oe[x]), lk[x]);
                }
        }
    }
    __device__ __forceinline__ double sweep(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int s, double2_t* xch) {
        request(a, rs, s);
        return finish(a, rs, s, xch);
    }
    __device__ __forceinline__ double finish(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int s, double2_t*) {
        const bool first = s == 0 && a.state_init == 0;
        const int cur = s & 1, nxt = cur ^ 1;
        const bool is_msg = tedge >= 0;
        DSTAMP_INPUTS(a, s);
#pragma unroll
        for (int x = 0; x < RC; ++x)
            if (x < dmax) {
#pragma unroll
                for (int i = 0; i < K; ++i) acc[i] *= (x != tpos) ? lk[x][i] : 1.0;  // x * 1.0 == x
            }
        for (int x0 = RC; x0 < dmax; x0 += RC) {  // nodes with more children than the registers hold: further round trips
            int ids[RC];
#pragma unroll
            for (int x = 0; x < RC; ++x) ids[x] = (x0 + x < deg) ? a.oedge[obeg + x0 + x] : 0;
#pragma unroll
            for (int x = 0; x < RC; ++x) {
#pragma unroll
                for (int i = 0; i < K; ++i) lk[x][i] = 1.0;
                if (!first && x0 + x < deg) dag_ld4(rs, dag_off_lam(a.E, a.n, cur, ids[x]), lk[x]);
            }
#pragma unroll
            for (int x = 0; x < RC; ++x)
#pragma unroll
                for (int i = 0; i < K; ++i) acc[i] *= (x0 + x != tpos) ? lk[x][i] : 1.0;
        }
        normalize_k<K>(acc);
        double wres = 0.0;
        if (active) {
            if (is_msg) {
#pragma unroll
                for (int i = 0; i < K; ++i) wres = res_acc(wres, fabs(acc[i] - old[i]));
                dag_st4(rs, dag_off_pim(a.E, a.n, nxt, tedge), acc);
            } else if (frozen) {
                dag_st4(rs, dag_off_nlam(a.E, a.n, nxt, snode), old);
            } else {
                dag_st4(rs, dag_off_nlam(a.E, a.n, nxt, snode), acc);
            }
        }
        return wres;
    }
};

// ---- the same with a node's c + 1 items in ADJACENT lanes of the wave (the planner's default: kDagParent).  Every lane loads ONE
// record -- the lambda(v) item pi(v), the item of the pi-message to child r that child's lambda-message -- and the node's lanes
// read each other's through the wave's scratch: 4 vector-memory instructions per lane and iteration whatever the child count.
struct DagParentX {
    static constexpr int K = 4;
    int node, snode, tedge, deg, tpos, first_lane, dmax, lane, kv;   // (tedge: a state record; snode: slot of the node's vectors)
    bool active, frozen;
    unsigned frozen_bits;   // bit q: the node is an evidence node of set q (frozen: of the set whose turn it is)
    __device__ __forceinline__ void turn(int q) { frozen = ((frozen_bits >> q) & 1u) != 0; }
    __device__ __forceinline__ void init(const DagArgs& a, const DagTile& t, int lane_) {
        const DagParentLaneDev it = a.pitem[t.lane_base + lane_];
        lane = lane_;
        active = it.node >= 0;
        node = active ? it.node : 0;
        snode = active ? it.snode : 0;
        tedge = active ? it.tedge : -1;
        deg = active ? (it.deg_tpos & 0xffff) : 0;
        tpos = active ? (it.deg_tpos >> 16) : -1;   // -1: the lambda(v) item, the first of its node's lanes
        first_lane = lane - (tpos + 1);
        dmax = t.dmax;
        kv = a.node_k ? a.node_k[node] : 4;
        frozen_bits = dag_frozen_bits(a, active, node); frozen = (frozen_bits & 1u) != 0;
    }
    // the inputs of the sweep in hand: requested by request(), consumed by finish() (see DagChildU)
    double rec[K], old[K];
    __device__ __forceinline__ void request(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int s) {
        const bool first = s == 0 && a.state_init == 0;   // (a padded network's initial state stands in memory)
        const int cur = s & 1;
        const bool is_msg = tedge >= 0;
#pragma unroll
        for (int i = 0; i < K; ++i) { rec[i] = 1.0; old[i] = 1.0; }
        if (is_msg) {
            if (!first) {
                dag_ld4(rs, dag_off_lam(a.E, a.n, cur, tedge), rec);   // the lambda-message of this item's own child
                dag_ld4(rs, dag_off_pim(a.E, a.n, cur, tedge), old);   // the previous pi-message, for the residual
            }
        } else {
            // pi(v): the previous sweep's, or in sweep 0 the initial one (1.0, a root's CPT row, the evidence)
            if (!first || frozen) dag_ld4(rs, dag_off_npi(a.E, a.n, cur, snode), rec);
            else if (active) {
#pragma unroll
                for (int i = 0; i < K; ++i) rec[i] = a.npi_init[int64_t(node) * 4 + i];
            }
            if (frozen) dag_ld4(rs, dag_off_nlam(a.E, a.n, cur, snode), old);  // an evidence node's lambda is carried over (:223)
        }
    }
    __device__ __forceinline__ double sweep(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int s, double2_t* xch) {
        request(a, rs, s);
        return finish(a, rs, s, xch);
    }
    __device__ __forceinline__ double finish(const DagArgs& a, __amdgpu_buffer_rsrc_t rs, int s, double2_t* xch) {
        const int nxt = (s & 1) ^ 1;
        const bool is_msg = tedge >= 0;
        double acc[K];
        double2_t r0, r1;
        r0.x = rec[0]; r0.y = rec[1]; r1.x = rec[2]; r1.y = rec[3];
        xch[2 * lane] = r0;
        xch[2 * lane + 1] = r1;
        lds_fence();
        DSTAMP_INPUTS(a, s);
        {
            const double2_t p0 = xch[2 * first_lane], p1 = xch[2 * first_lane + 1];
            // (lambda(v): the product over the children starts from ones over the node's OWN states -- a leaf's stays there)
            acc[0] = is_msg ? p0.x : 1.0; acc[1] = is_msg ? p0.y : (kv > 1 ? 1.0 : 0.0); acc[2] = is_msg ? p1.x : (kv > 2 ? 1.0 : 0.0); acc[3] = is_msg ? p1.y : (kv > 3 ? 1.0 : 0.0);
        }
        for (int x = 0; x < dmax; ++x) {   // ascending children, the target left out (:207-214, :229-235); wave-uniform trip count
            const int src = (x < deg) ? first_lane + 1 + x : lane;
            const double2_t l0 = xch[2 * src], l1 = xch[2 * src + 1];
            const bool use = x < deg && x != tpos;
            acc[0] *= use ? l0.x : 1.0;   // x * 1.0 == x
            acc[1] *= use ? l0.y : 1.0;
            acc[2] *= use ? l1.x : 1.0;
            acc[3] *= use ? l1.y : 1.0;
        }
        lds_fence();
        normalize_k<K>(acc);
        double wres = 0.0;
        if (active) {
            if (is_msg) {
#pragma unroll
                for (int i = 0; i < K; ++i) wres = res_acc(wres, fabs(acc[i] - old[i]));
                dag_st4(rs, dag_off_pim(a.E, a.n, nxt, tedge), acc);
            } else if (frozen) {
                dag_st4(rs, dag_off_nlam(a.E, a.n, nxt, snode), old);
            } else {
                dag_st4(rs, dag_off_nlam(a.E, a.n, nxt, snode), acc);
            }
        }
        return wres;
    }
};

// one tile, whatever its kind: state set up, f(state) called
template <class F>
__device__ __forceinline__ void dag_with_tile(const DagArgs& a, const DagTile& t, int lane, F&& f) {
    switch (t.kind) {
        case 0: { DagChildU<0> st; st.init(a, t, lane); f(st); break; }
        case 1: { DagChildU<1> st; st.init(a, t, lane); f(st); break; }
        case 2: { DagChildU<2> st; st.init(a, t, lane); f(st); break; }
        case 3: { DagChildG<1> st; st.init(a, t, lane); f(st); break; }
        case 4: { DagChildG<2> st; st.init(a, t, lane); f(st); break; }
        case 5: { DagChildG<3> st; st.init(a, t, lane); f(st); break; }
        case kDagParent: { DagParentX st; st.init(a, t, lane); f(st); break; }
        default: { DagParent st; st.init(a, t, lane); f(st); break; }
    }
}
template <class T> struct dag_has_belief { static constexpr bool value = true; };
template <> struct dag_has_belief<DagParent> { static constexpr bool value = false; };
template <> struct dag_has_belief<DagParentX> { static constexpr bool value = false; };

// STREAM = false: at most one tile per wave, its static state (CPT, ids, marks) in registers for the whole run.
// STREAM = true: a wave walks its tiles [slot_ptr[slot], slot_ptr[slot + 1]) every iteration, setting each up again.
// FLOW (single query, one tile per wave): the dataflow form -- no grid barrier, one more block serves the stop decision.
template <bool STREAM, bool BATCH, bool FLOW = false>
__global__ __launch_bounds__(kDagWaves * kWave) void bp_dag_kernel(DagArgs a) {
    static_assert(!FLOW || (!STREAM && !BATCH), "the dataflow form runs one evidence set at one tile per wave");
    __shared__ DagShared sh;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (FLOW) {
        if (int(blockIdx.x) == a.n_blocks) { dag_flow_service(a, sh, lane, wave); return; }
        const int slot_f = blockIdx.x * kDagWaves + wave;
        const int tf0 = a.slot_ptr[slot_f], tf1 = a.slot_ptr[slot_f + 1];
        if (tf1 <= tf0) return;   // (a wave without a tile has no part in the protocol)
        const __amdgpu_buffer_rsrc_t rsf = __builtin_amdgcn_make_buffer_rsrc(a.state, 0, int(dag_state_doubles(a.E, a.n) * 8), 0x00020000);
        const DagTile tdf = a.tiles[tf0];
        dag_with_tile(a, tdf, lane, [&](auto& st) {
            (void)dag_flow_drive(a, tf0, lane, wave, [&](int, int s) { st.request(a, rsf, s); },
                                 [&](int, int s) { return st.finish(a, rsf, s, sh.xch[wave]); },
                                 [&](int, int n, int done) {
                                     if constexpr (dag_has_belief<std::remove_reference_t<decltype(st)>>::value) {
                                         if (done != 0) st.belief(a, rsf, n, a.b.beliefs);
                                     }
                                 });
        });
        return;   // (a wave that gave up has raised the abort word and the host's flag itself; the service block reports)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long now = wall_clock64();
        for (int q = 0; q < a.n_sets; ++q) a.host_ctl[q].t_first = now;
    }
    if (threadIdx.x < kDagMaxSets) {
        DagSetShared& z = sh.set[threadIdx.x];
        z.skew_ticks = 0; z.t_arrive = 0; z.arrived = 0; z.pub_it = -1; z.poll_it = -1; z.ver = -1;
    }
    if constexpr (BATCH) __syncthreads();   // (a single query reads these words behind its first block barrier only)
    const int slot = blockIdx.x * kDagWaves + wave;
    const int t0 = a.slot_ptr[slot], t1 = a.slot_ptr[slot + 1];
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(a.state, 0, int(dag_state_doubles(a.E, a.n) * 8), 0x00020000);
    auto rs_of = [&](int q) {   // set q's state
        if constexpr (!BATCH) return rs0;
        else return __builtin_amdgcn_make_buffer_rsrc(a.state + int64_t(q) * a.state_stride, 0, int(dag_state_doubles(a.E, a.n) * 8), 0x00020000);
    };
    auto beliefs_of = [&](int q) { return a.b.beliefs + int64_t(q) * a.belief_stride; };
    bool ok = true;
    if (t1 <= t0) {
        ok = dag_drive<BATCH>(a, sh, lane, wave, [](int, int) {}, [](int, int) { return 0.0; }, [](int, int, int) {});
    } else if constexpr (STREAM) {
        ok = dag_drive<BATCH>(a, sh, lane, wave, [](int, int) {},
                       [&](int q, int s) {
                           double w = 0.0;
                           const __amdgpu_buffer_rsrc_t rs = rs_of(q);
                           for (int t = t0; t < t1; ++t) {
                               const DagTile td = a.tiles[t];
                               dag_with_tile(a, td, lane, [&](auto& st) { if constexpr (BATCH) st.turn(q); w = res_acc(w, st.sweep(a, rs, s, sh.xch[wave])); });
                           }
                           return w;
                       },
                       [&](int q, int n, int done) {
                           if (done == 0) return;
                           const __amdgpu_buffer_rsrc_t rs = rs_of(q);
                           for (int t = t0; t < t1; ++t) {
                               const DagTile td = a.tiles[t];
                               if (td.kind >= kDagParent) continue;
                               dag_with_tile(a, td, lane, [&](auto& st) {
                                   if constexpr (dag_has_belief<std::remove_reference_t<decltype(st)>>::value) st.belief(a, rs, n, beliefs_of(q));
                               });
                           }
                       });
    } else {
        const DagTile td = a.tiles[t0];
        dag_with_tile(a, td, lane, [&](auto& st) {
            ok = dag_drive<BATCH>(a, sh, lane, wave, [&](int q, int s) { if constexpr (BATCH) st.turn(q); st.request(a, rs_of(q), s); },
                           [&](int q, int s) { return st.finish(a, rs_of(q), s, sh.xch[wave]); },
                           [&](int q, int n, int done) {
                               if constexpr (dag_has_belief<std::remove_reference_t<decltype(st)>>::value) {
                                   if (done != 0) st.belief(a, rs_of(q), n, beliefs_of(q));
                               }
                           });
        });
    }
    // a block that gave up a bounded wait says so itself: block 0 may long have reported its own outcome
    if (!ok && threadIdx.x == 0 && a.host_abort) __hip_atomic_store(a.host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long now = wall_clock64();
        for (int q = 0; q < a.n_sets; ++q) {
            a.host_ctl[q].t_last = now;
            if (!ok) { a.host_ctl[q].run_id = a.run_id; a.host_ctl[q].done = -1; }
        }
    }
}

// bn_bp_set_evidence for this path (:68-73): pi(v) = lambda(v) = the given vector in BOTH buffers (an observed node's vectors
// are carried over by every sweep), node marked with this set's mark value (the previous set's marks need no clearing)
__device__ __forceinline__ void dag_apply_evidence(const DagEvidenceArgs& a) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.ne) return;
    const int v = a.ev_node[j];
    const int sv = a.nperm[v];   // the slot of its vectors
    const int kv = a.node_k ? a.node_k[v] : 4;
    for (int i = 0; i < 4; ++i) {
        const double x = i < kv ? a.ev_val[a.ev_off[j] + i] : 0.0;
        for (int par = 0; par < 2; ++par) {
            a.state[dag_off_npi(a.E, a.n, par, sv) * 2 + i] = x;
            a.state[dag_off_nlam(a.E, a.n, par, sv) * 2 + i] = x;
        }
    }
    a.frz[v] = a.frz_mark;
}
__global__ __launch_bounds__(256) void dag_evidence_kernel(DagEvidenceArgs a) { dag_apply_evidence(a); }
__global__ __launch_bounds__(256) void dag_evidence_batch_kernel(DagEvidenceBatch b) { dag_apply_evidence(b.set[blockIdx.y]); }

__device__ __forceinline__ void dag_apply_init(const DagInitArgs& a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < a.E) {   // both messages of edge t: ones over the states of its PARENT
        const int kp = a.node_k[a.in_idx[t]];
        const int rec = a.eperm[t];
        for (int i = 0; i < 4; ++i) {
            const double x = i < kp ? 1.0 : 0.0;
            a.state[dag_off_pim(a.E, a.n, 0, rec) * 2 + i] = x;
            a.state[dag_off_lam(a.E, a.n, 0, rec) * 2 + i] = x;
        }
    }
    if (t < a.n && a.frz[t] != a.frz_mark) {
        const int kv = a.node_k[t];
        const int st = a.nperm[t];
        for (int i = 0; i < 4; ++i) {
            a.state[dag_off_npi(a.E, a.n, 0, st) * 2 + i] = a.npi_init[int64_t(t) * 4 + i];
            a.state[dag_off_nlam(a.E, a.n, 0, st) * 2 + i] = i < kv ? 1.0 : 0.0;
        }
    }
}
__global__ __launch_bounds__(256) void dag_init_kernel(DagInitArgs a) { dag_apply_init(a); }
__global__ __launch_bounds__(256) void dag_init_batch_kernel(DagInitBatch b) { dag_apply_init(b.set[blockIdx.y]); }
int launch_dag_init_batch(const DagInitBatch& b, int n_sets, void* stream_handle) {
    (void)hipGetLastError();
    const int work = b.set[0].E > b.set[0].n ? b.set[0].E : b.set[0].n;   // (one network: the same for every set)
    if (work <= 0 || n_sets <= 0) return 0;
    hipLaunchKernelGGL(dag_init_batch_kernel, dim3((work + 255) / 256, n_sets), dim3(256), 0, (hipStream_t)stream_handle, b);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}
int launch_dag_evidence_batch(const DagEvidenceBatch& b, int n_sets, void* stream_handle) {
    (void)hipGetLastError();
    int ne_max = 0;
    for (int q = 0; q < n_sets; ++q) ne_max = b.set[q].ne > ne_max ? b.set[q].ne : ne_max;
    if (ne_max <= 0) return 0;
    hipLaunchKernelGGL(dag_evidence_batch_kernel, dim3((ne_max + 255) / 256, n_sets), dim3(256), 0, (hipStream_t)stream_handle, b);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}
int launch_dag_init(const DagInitArgs& a, void* stream_handle) {
    (void)hipGetLastError();
    const int work = a.E > a.n ? a.E : a.n;
    if (work <= 0) return 0;
    hipLaunchKernelGGL(dag_init_kernel, dim3((work + 255) / 256), dim3(256), 0, (hipStream_t)stream_handle, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}
int launch_bp_dag(const DagArgs& a, bool stream, void* stream_handle) {
    (void)hipGetLastError();
    const bool batch = a.n_sets > 1;
    const bool flow = a.flow != nullptr && !stream && !batch && a.n_blocks > 1;
    const dim3 g(a.n_blocks + (flow ? 1 : 0)), t(kDagWaves * kWave);
    if (flow) hipLaunchKernelGGL((bp_dag_kernel<false, false, true>), g, t, 0, (hipStream_t)stream_handle, a);
    else if (stream && batch) hipLaunchKernelGGL((bp_dag_kernel<true, true>), g, t, 0, (hipStream_t)stream_handle, a);
    else if (stream) hipLaunchKernelGGL((bp_dag_kernel<true, false>), g, t, 0, (hipStream_t)stream_handle, a);
    else if (batch) hipLaunchKernelGGL((bp_dag_kernel<false, true>), g, t, 0, (hipStream_t)stream_handle, a);
    else hipLaunchKernelGGL((bp_dag_kernel<false, false>), g, t, 0, (hipStream_t)stream_handle, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}
int launch_dag_evidence(const DagEvidenceArgs& a, void* stream_handle) {
    (void)hipGetLastError();
    if (a.ne <= 0) return 0;
    hipLaunchKernelGGL(dag_evidence_kernel, dim3((a.ne + 255) / 256), dim3(256), 0, (hipStream_t)stream_handle, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

}  // namespace bnmi

#ifdef BN_TILE_CLOCK
extern "C" int bn_debug_dag_clock(unsigned long long* out, int n_waves) {
    if (n_waves > bnmi::kDagMaxBlocks * bnmi::kDagWaves) n_waves = bnmi::kDagMaxBlocks * bnmi::kDagWaves;
    return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(bnmi::g_dag_clock), sizeof(unsigned long long) * 12 * n_waves));
}
#endif

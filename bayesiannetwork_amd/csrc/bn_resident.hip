// bn_resident.hip -- the whole belief-propagation run in ONE launch, tiles resident on the chip.
//
// For networks made of register-resident tiles only (uniform arity k in {2,3,4}, CPT <= 64 entries:
// every grid, chain and tree of the reference's tests, BASELINE.json configs[2]) and small enough that
// every tile gets its own wavefront (<= 8 waves x 256 CUs).  Each wave keeps, for the whole run,
//   * its tile's CPT image in VGPRs (loaded once: 57 % of a sweep's bytes on the 316x316 grid),
//   * its out-edge references and evidence marks,
//   * pi(v) / lambda(v) of its nodes (never written to memory before the run ends),
// so one iteration of the reference's while(true) loop (belief_propagation.hpp:75-148) moves the
// messages only: incoming messages read, outgoing messages written.  The arithmetic is tile_uniform's
// (bn_tiles.hpp), statement for statement: results are bit-identical to the launch path (asserted).
//
// Between iterations stands ONE grid barrier (the Jacobi schedule needs nothing finer):
//   producer  message stores are 16-byte WRITE-THROUGH (sc1) stores; every wave drains them
//             (s_waitcnt vmcnt(0)), __syncthreads(), then lane 0 of the block arrives;
//   barrier   atomic-free: a tile block publishes a pair of 8-byte {generation, residual half} granules.  One evidence set
//             (ResidentArgs::direct, the default): the first wave of EVERY tile block sweeps all blocks' granules itself --
//             one memory round trip per sweep (sweep_granules) -- reduces the residual and takes the same decision.
//             Several sets per launch, or "direct" 0: ONE SERVICE block (an extra block of the launch on a CU the tiles
//             leave free) does the sweep and publishes generation + verdict in one word per group of tile blocks
//             (blockIdx % 8: spreads the pollers), which the tile blocks poll (relaxed agent-scope loads + s_sleep).
//             Loads of the records are sc1, so no acquire fence is needed.  (cdna_hip_programming.md Guideline 16 R2 /
//             MI355X_MICROARCH.md barrier-xcd, handoff-1to1.)  Nothing depends on dispatch order or on which XCD a block runs.
//   residual  max|new - old| (:105-131) travels in the granules; whoever collects them takes the stop decision
//             (:147): every block arrives at the same verdict.
// A one-block grid (<= 8 tiles: Pearl's network, small chains) needs no atomics at all: LDS slots and
// __syncthreads().
//
// SEVERAL EVIDENCE SETS PER LAUNCH (bn_bp_run_batch).  The CPT image, the references and the layout are the
// same for every query on a network; only messages, node vectors and evidence marks are per query.  The
// kernel walks B <= kResidentMaxSets = 4 sets round-robin -- sweep s of set A, arrive at A's barrier, sweep s of set B, arrive at B's,
// wait for A's barrier, sweep s+1 of A, ... -- so the ~3.5 us a barrier takes to complete are spent computing
// the other sets instead of waiting, and one resident CPT serves all of them.  Every set has its own
// record / node buffers, barrier words, residuals and verdict, and leaves the rotation on the sweep ITS
// reference run would stop on; node vectors of a set go through memory between its turns (no registers left
// for B copies).  Results per set are bit-identical to a run of that set alone (asserted).
//
// Every wait is bounded (100 MHz wall clock); a wait that gives up raises `abort`, every block leaves,
// and the host redoes the run with per-sweep launches.  A launch executes at most `budget` iterations;
// a run that needs more continues on the launch path from the state this kernel leaves in memory.
#include "bn_tiles.hpp"

namespace bnmi {

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Diagnostic builds (-DBN_TILE_CLOCK): lane 0 of every wave stamps the phases of ONE iteration (the sixth) of the run
#ifdef BN_TILE_CLOCK
#define RSTAMP(k, iter, lane_, wave_)                                                              \
    do {                                                                                           \
        if ((iter) == 5 && (lane_) == 0) g_tile_clock[blockIdx.x * kResidentWaves + (wave_)][k] = wall_clock64(); \
    } while (0)
#else
#define RSTAMP(k, iter, lane_, wave_) ((void)0)
#endif

// Message records are exchanged between CUs / XCDs inside the launch: every record access is a
// 16-byte sc1 access through a buffer descriptor (aux 16 = sc1) -- stores write through, loads
// bypass the CU's L1 -- the form of Guideline 16 that needs neither a release nor an acquire fence
// around the barrier (an agent-scope acquire costs ~1.7 us per block and sweep).  They are ordinary
// compiler-tracked memory operations; offsets are 32-bit (the host admits record buffers < 2 GiB).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2_t ld_rec(__amdgpu_buffer_rsrc_t r, int64_t idx2) {
    return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(r, int(idx2) * 16, 0, 16));
}
__device__ __forceinline__ void st_rec(__amdgpu_buffer_rsrc_t r, int64_t idx2, double2_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, int(idx2) * 16, 0, 16);
}
// the same at system scope (aux 17 = sc0 sc1): cut-edge halves, read from this rank's exchange region where a peer
// stored them, or stored into the peer's
__device__ __forceinline__ double2_t ld_rec_sys(__amdgpu_buffer_rsrc_t r, int64_t idx2) {
    return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(r, int(idx2) * 16, 0, 17));
}
__device__ __forceinline__ void st_rec_sys(__amdgpu_buffer_rsrc_t r, int64_t idx2, double2_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, int(idx2) * 16, 0, 17);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Pins a value at this point of the program: the empty asm is opaque to every compiler pass, so nothing
// that produces `x` can sink below it and nothing that consumes it can rise above.  Used to keep the
// unrolled CPT contraction in own-state order -- left alone, instruction selection interleaves all
// K * C products of the unrolled loops and needs far more than 256 registers for their temporaries.
__device__ __forceinline__ void pin_here(double& x) { asm volatile("" : "+v"(x)); }

struct BlockShared {
    unsigned long long slot[kResidentMaxSets][kResidentWaves];  // per-set, per-wave residual bit patterns
    int verdict[kResidentMaxSets];                              // of a set's last barrier: kGoOn / kConverged / kCapped / kAbort
    unsigned long long t_arrive;                                // direct form: 100 MHz clock when this block published its granules
    int skew_ticks;                                             // ... and how long after this block the LAST block arrived in the previous iteration
    // several sets per launch: the block's waves are not held together by block barriers (arrive_batch / wait_verdict_batch)
    int arrived[kResidentMaxSets];                              // waves that have finished the set's sweep in hand (the last one publishes)
    int poll_it[kResidentMaxSets];                              // the last iteration for whose barrier a wave of the block has taken the polling on
    int ver[kResidentMaxSets];                                  // the last iteration whose verdict is out, with the verdict: iteration << 2 | verdict
};
enum : int { kGoOn = 0, kConverged = 1, kCapped = 2, kAbort = 3 };

// maximum_difference (:105) from its bit pattern (non-negative doubles order like unsigned integers)
__device__ __forceinline__ double residual_of(unsigned long long bits) {
    const double r = __longlong_as_double((long long)bits);
    return r < DBL_MIN ? DBL_MIN : r;  // it starts at numeric_limits<double>::min()
}
__device__ __forceinline__ int verdict_of(const ResidentArgs& a, double r, int n_done) {
    if (r < a.eps) return kConverged;                             // strict '<' (:147)
    if (a.max_sweeps > 0 && n_done >= a.max_sweeps) return kCapped;
    return kGoOn;
}

// ---- the grid barrier -------------------------------------------------------------------------
// The tile blocks never wait inside an arrival and never execute a returning atomic.  A block that has
// finished a sweep of a set publishes ONE pair of 8-byte granules {generation | half of its residual's bit
// pattern} (cdna_hip_programming.md Guideline 16, form R2: the data is the flag, one aligned 8-byte sc1
// store each, nothing to order).
// The granules are collected by a SERVICE block: one extra block of the same launch that carries no tile and
// sits on a CU the tiles leave free (the host admits the path only when tile blocks + 1 fit the chip): it
// sweeps the granules of every tile block until all carry the generation, reduces the residual, decides
// (converged / capped / go on), records the residual and publishes generation + verdict in one word per
// group of tile blocks (blockIdx % 8: spreads the pollers), which the tile blocks poll when they next need
// that set.  Generations count on from launch to launch (gen_base), so nothing has to be zeroed between
// launches: a stale granule or word always carries a smaller generation.
// Round trips on the path from the last tile block's arrival to the verdict: granule store ->
// sweep -> verdict store -> tile blocks' poll; none of them on a
// tile block.  With several sets in flight the whole of it runs behind the other sets' sweeps.
// (A two-level collection -- one service block per group, then a top block -- was measured ~0.3 us slower per
// barrier than this single sweep over all tile blocks' granules.)
// Direct form (ResidentArgs::direct, one evidence set -- the default there since round 3): no service block; the first wave of
// EVERY tile block sweeps all blocks' granules itself and takes the decision (wait_verdict): one hand-off per barrier instead
// of two, 0.3-0.5 us per sweep (32 x 32 grid 6.98 -> 6.46 us, 316 x 316 11.46 -> 11.04).  Granules of consecutive iterations
// alternate between two tables, so a block already past a barrier cannot overwrite what a slower block still has to read.
__device__ __forceinline__ unsigned long long granule(unsigned gen, unsigned half) { return ((unsigned long long)gen << 32) | half; }

// First half of the barrier of iteration `it` (sweep s) of evidence set `set`: publish the block's residual.
__device__ __forceinline__ void arrive(const ResidentArgs& a, BlockShared& sh, int set, int it, int s, double wres, int lane,
                                       int wave) {
    const unsigned long long bits = wave_umax((unsigned long long)__double_as_longlong(wres));
    if (lane == 0) sh.slot[set][wave] = bits;
    RSTAMP(3, it, lane, wave);  // sweep issued
    drain_stores();  // this wave's write-through stores have reached memory
    RSTAMP(4, it, lane, wave);  // stores drained
    __syncthreads();
    RSTAMP(5, it, lane, wave);  // the block's waves have all arrived
    if (threadIdx.x == 0) {
        unsigned long long m = 0;
        for (int w = 0; w < a.waves; ++w) m = sh.slot[set][w] > m ? sh.slot[set][w] : m;
        ResidentSync* sy = a.sync + set;
        if (a.n_tile_blocks == 1) {
            sy->res[it] = m;
            sh.verdict[set] = verdict_of(a, residual_of(m), s + 1);
        } else {
            const unsigned gen = a.gen_base + unsigned(it) + 1u;
            // direct form: the granules of consecutive iterations alternate between two tables -- a block that is already past
            // this barrier must not overwrite what a slower block still has to read
            unsigned long long* g = (a.direct != 0 && (it & 1)) ? sy->blk_odd[blockIdx.x] : sy->blk[blockIdx.x];
            // Second granule: {arrival time: 16 bits of the 100 MHz clock | low 16 bits of the generation (tells a torn pair: a slot
            // is reused every second generation at the earliest) | residual low half}.  The work per iteration is static, so the
            // arrival times of one iteration predict when the LAST block arrives in the next: a collecting wave places its first
            // poll there instead of polling from its own arrival on -- a poll is a ~1 us round trip, and one that leaves just before
            // the last granule becomes visible costs the block a whole second trip (round 4).
            const unsigned long long now = wall_clock64();
            sh.t_arrive = now;
            __hip_atomic_store(g, granule(gen, unsigned(m >> 32)), RLX_AGENT);
            __hip_atomic_store(g + 1, ((now & 0xffffull) << 48) | ((unsigned long long)(gen & 0xffffu) << 32) | unsigned(m), RLX_AGENT);
        }
    }
}

// Several sets per launch (service-block form): no block barrier.  A wave that has finished its sweep of a set counts itself in and goes
// on to the next set's sweep; the wave that completes the count publishes the block's granules (bn_dag.hip dag_arrive: with a
// __syncthreads here and another behind the verdict every set-turn cost the block its slowest wave twice -- stamps, 316 x 316 grid, four
// sets: 0.4 us median / 1.6 us at the 90th percentile in this barrier, 0.8 us in the verdict's).
__device__ __forceinline__ void arrive_batch(const ResidentArgs& a, BlockShared& sh, int set, int it, int s, double wres, int lane, int wave) {
    const unsigned long long bits = wave_umax((unsigned long long)__double_as_longlong(wres));
    if (lane == 0) sh.slot[set][wave] = bits;
    RSTAMP(3, it, lane, wave);
    drain_stores();  // this wave's write-through stores have reached memory
    RSTAMP(4, it, lane, wave);
    if (lane == 0) {
        const int before = __hip_atomic_fetch_add(&sh.arrived[set], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (before == a.waves - 1) {   // the block's last wave: every wave's stores are out, every slot is written
            __hip_atomic_store(&sh.arrived[set], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (next used by a wave that has seen this barrier's verdict)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            unsigned long long m = 0;
            for (int w = 0; w < a.waves; ++w) m = sh.slot[set][w] > m ? sh.slot[set][w] : m;
            ResidentSync* sy = a.sync + set;
            if (a.n_tile_blocks == 1) {
                sy->res[it] = m;
                __hip_atomic_store(&sh.ver[set], it << 2 | verdict_of(a, residual_of(m), s + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                const unsigned gen = a.gen_base + unsigned(it) + 1u;
                unsigned long long* g = sy->blk[blockIdx.x];   // (the service block reads one table: it is never behind a tile block's next arrival)
                const unsigned long long now = wall_clock64();
                __hip_atomic_store(g, granule(gen, unsigned(m >> 32)), RLX_AGENT);
                __hip_atomic_store(g + 1, ((now & 0xffffull) << 48) | ((unsigned long long)(gen & 0xffffu) << 32) | unsigned(m), RLX_AGENT);   // (arrive()'s format)
            }
        }
    }
    RSTAMP(5, it, lane, wave);
}

// polls until pred() or abort / timeout; false = give up (abort raised)
template <class Pred>
__device__ __forceinline__ bool poll_until(const ResidentArgs& a, Pred&& pred) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned n = 1;; ++n) {
        if (pred()) return true;
        // the abort word is a memory round trip of its own and the clock a scalar-memory one: looked at every 8th poll only (a poll
        // is ~1 us; the bound is 50 ms), so that the period of the poll -- what the last arrival waits for -- is ONE round trip
        if ((n & 7u) == 0) {
            if (__hip_atomic_load(&a.sync->abort, RLX_AGENT) != 0) return false;  // set 0's word is the launch's abort flag
            if (wall_clock64() - t0 > a.timeout_ticks) {
                if (pred()) return true;  // a wave that was descheduled across the deadline looks once more before giving up
                __hip_atomic_store(&a.sync->abort, 1u, RLX_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// One sweep over the granule pairs of all tile blocks in ONE memory round trip: lane l requests the pairs of blocks l, l + 64,
// l + 128, l + 192 back to back as 16-byte sc1 loads (table base in SGPRs + lane offset + immediate) and awaits them together
// -- as a loop of dependent 8-byte loads a sweep over the 316x316 grid's 196 blocks took four round trips, and the last block
// to arrive is seen one whole sweep later (headline query 0.139 -> 0.130 ms).  Generation and value come with the same load; a
// pair may be torn between its halves, each of which carries its own generation.  Slots of blocks >= nb lie inside the table,
// are never written and are ignored.  Returns whether every block of this lane carries `gen`; acc = maximum of their values.
__device__ __forceinline__ bool sweep_granules(const unsigned long long* tbl, int lane, int nb, unsigned gen, unsigned long long& acc,
                                               unsigned own16 = 0u, int* late_out = nullptr) {
    static_assert(kResidentMaxBlocks == 4 * kWave, "four pairs per lane cover the table");
    const unsigned voff = unsigned(lane) * 16u;
    u32x4 r0, r1, r2, r3;
    if (nb <= kWave)
        asm volatile("global_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r0) : "v"(voff), "s"(tbl) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %4, %5 sc1\n\t"
                     "global_load_dwordx4 %1, %4, %5 offset:1024 sc1\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:2048 sc1\n\t"
                     "global_load_dwordx4 %3, %4, %5 offset:3072 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(voff), "s"(tbl) : "memory");
    bool mine = true;
    acc = 0;
    int late = 0;
    auto take = [&](const u32x4& r, int blk) {  // words: {residual high half, generation, residual low half, arrival time << 16 | generation & 0xffff}
        if (blk < nb) {
            mine = mine && r.y == gen && (r.w & 0xffffu) == (gen & 0xffffu);
            const unsigned long long v = (unsigned long long)r.x << 32 | r.z;
            acc = v > acc ? v : acc;
            const int d = int(short((r.w >> 16) - own16));   // that block's arrival after this one's, 10 ns ticks
            late = d > late ? d : late;
        }
    };
    take(r0, lane);
    if (nb > kWave) { take(r1, lane + kWave); take(r2, lane + 2 * kWave); take(r3, lane + 3 * kWave); }
    if (late_out) *late_out = late;
    return mine;
}

// Second half: the verdict of iteration `it` of `set` once every block has arrived.
__device__ __forceinline__ int wait_verdict(const ResidentArgs& a, BlockShared& sh, int set, int it) {
    if (a.direct != 0 && a.n_tile_blocks > 1) {
        // Direct form (one evidence set): the first wave of EVERY tile block sweeps all blocks' granules itself -- lane l those of
        // blocks l, l + 64, ... -- reduces the residual and takes the (same) decision: one hand-off (granule store -> the other
        // blocks' poll) instead of two (granule -> service block -> verdict word -> poll).
        if (threadIdx.x < kWave) {
            int lane = int(threadIdx.x);
            // opaque to the optimiser: otherwise the per-lane granule addresses are hoisted out of the sweep loop as loop invariants
            // and, with the tile's CPT and vectors live in this wave (256 VGPRs), spilled to scratch and reloaded every sweep
            asm volatile("" : "+v"(lane));
            ResidentSync* sy = a.sync + set;
            const unsigned gen = a.gen_base + unsigned(it) + 1u;
            const int nb = a.n_tile_blocks;
            const unsigned long long* tbl = (it & 1) ? &sy->blk_odd[0][0] : &sy->blk[0][0];
            unsigned long long m = 0;
            // first poll: when the last block is expected to arrive (this block's arrival + the previous iteration's skew) + a margin
            const unsigned long long t_arr = sh.t_arrive;
            const unsigned own16 = unsigned(t_arr) & 0xffffu;
            if (a.first_poll_delay >= 0) {   // (negative: poll from this block's own arrival on)
                const unsigned long long first_at = t_arr + (unsigned long long)(sh.skew_ticks + a.first_poll_delay);
                while (wall_clock64() < first_at) __builtin_amdgcn_s_sleep(1);
            }
            int late = 0;
            const bool ok = poll_until(a, [&] { return __all(sweep_granules(tbl, lane, nb, gen, m, own16, &late)) != 0; });
            m = wave_umax(m);
            {
                int mx = late;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(mx, off, kWave); mx = o > mx ? o : mx; }
                if (lane == 0) sh.skew_ticks = ok ? (mx < 400 ? mx : 400) : 0;
            }
            if (lane == 0) {
                if (blockIdx.x == 0 && ok) __hip_atomic_store(&sy->res[it], m, RLX_AGENT);   // the residual history (read back by this block)
                sh.verdict[set] = ok ? verdict_of(a, residual_of(m), a.sweep_begin + it + 1) : kAbort;
            }
        }
        __syncthreads();
        return sh.verdict[set];
    }
    if (threadIdx.x == 0 && a.n_tile_blocks > 1) {
        ResidentSync* sy = a.sync + set;
        const unsigned gen = a.gen_base + unsigned(it) + 1u;
        const int groups = a.n_tile_blocks < 8 ? a.n_tile_blocks : 8;
        const int g = blockIdx.x % groups;
        unsigned word = 0;
        const bool ok = poll_until(a, [&] {
            word = __hip_atomic_load(&sy->grp[g].gen, RLX_AGENT);
            return (word & 0x3fffffffu) >= gen;
        });
        sh.verdict[set] = ok ? int(word >> 30) : kAbort;
    }
    __syncthreads();
    return sh.verdict[set];
}

// ... and its verdict: the first wave of the block to need it does the polling (an LDS fetch-max decides who) and puts it out in LDS,
// the others wait for it there (bn_dag.hip dag_wait).
__device__ __forceinline__ int wait_verdict_batch(const ResidentArgs& a, BlockShared& sh, int set, int it, int lane) {
    int out = __hip_atomic_load(&sh.ver[set], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((out >> 2) == it) return out & 3;
    int before = it;
    if (lane == 0) before = __hip_atomic_fetch_max(&sh.poll_it[set], it, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (__builtin_amdgcn_readfirstlane(before) < it && a.n_tile_blocks > 1) {
        if (lane == 0) {
            ResidentSync* sy = a.sync + set;
            const unsigned gen = a.gen_base + unsigned(it) + 1u;
            const int groups = a.n_tile_blocks < 8 ? a.n_tile_blocks : 8;
            const int g = blockIdx.x % groups;
            unsigned word = 0;
            const bool ok = poll_until(a, [&] {
                word = __hip_atomic_load(&sy->grp[g].gen, RLX_AGENT);
                return (word & 0x3fffffffu) >= gen;
            });
            __hip_atomic_store(&sh.ver[set], it << 2 | (ok ? int(word >> 30) : kAbort), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    const unsigned long long t0 = wall_clock64();
    for (unsigned n = 1; ((out = __hip_atomic_load(&sh.ver[set], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 2) != it; ++n) {
        if ((n & 63u) == 0 && wall_clock64() - t0 > 2 * a.timeout_ticks) {   // (the poller gives up first and says so here)
            __hip_atomic_store(&a.sync->abort, 1u, RLX_AGENT);
            return kAbort;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return out & 3;
}

// The service block (one wave; the launch has exactly one when it has more than one tile block): lane l
// sweeps the granule pairs of tile blocks l, l + 64, l + 128, l + 192 until all carry the generation, the wave
// reduces the residual, lane 0 decides and publishes.  It follows the same (iteration, set) order as the tile
// blocks and knows from its own verdicts which sets are still running.
__device__ __forceinline__ void resident_service(const ResidentArgs& a, int lane) {
    if (a.direct != 0) return;  // every tile block collects the granules itself
    const int nb = a.n_tile_blocks;
    const int groups = nb < 8 ? nb : 8;
    unsigned active = a.set_mask;
    for (int it = 0; it < a.budget; ++it) {
        const unsigned gen = a.gen_base + unsigned(it) + 1u;
        for (int set = 0; set < a.n_sets; ++set) {
            if (((active >> set) & 1u) == 0) continue;
            ResidentSync* sy = a.sync + set;
            unsigned long long m = 0;
            if (!poll_until(a, [&] { return __all(sweep_granules(&sy->blk[0][0], lane, nb, gen, m)) != 0; }))
                return;
            m = wave_umax(m);
            const int verdict = verdict_of(a, residual_of(m), a.sweep_begin + it + 1);
            if (lane == 0) {
                __hip_atomic_store(&sy->res[it], m, RLX_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the residual is recorded before anyone learns the verdict
                const unsigned word = gen | (unsigned(verdict) << 30);
                for (int q = 0; q < groups; ++q) __hip_atomic_store(&sy->grp[q].gen, word, RLX_AGENT);
            }
            if (verdict != kGoOn) active &= ~(1u << set);
        }
        if (active == 0) break;
    }
}

// The launch's loop over iterations and evidence sets, shared by every tile kind.
//   phase(set, s)          one sweep of this wave's tile for `set`; returns the wave's residual contribution
//   finalize(set, n, done) the set stopped after n sweeps (done = kConverged / kCapped) or the launch's budget
//                          ran out (done = 0): node vectors to memory where needed, beliefs when done
// Returns false when a bounded wait gave up.
template <bool BATCH, class Phase, class Finalize>
__device__ __forceinline__ bool resident_drive(const ResidentArgs& a, BlockShared& sh, int lane, int wave, Phase&& phase,
                                               Finalize&& finalize) {
    unsigned active = BATCH ? a.set_mask : 1u;
    const int n_sets = BATCH ? a.n_sets : 1;  // one set: everything about sets folds away at compile time
    for (int it = 0; it <= a.budget; ++it) {  // the pass it == budget only collects verdicts
        const int s = a.sweep_begin + it;
        for (int set = 0; set < n_sets; ++set) {
            if (((active >> set) & 1u) == 0) continue;
            int v = kGoOn;
            RSTAMP(0, it, lane, wave);
            if (it > 0) {
                if (BATCH && a.direct == 0) v = wait_verdict_batch(a, sh, set, it - 1, lane);
                else v = wait_verdict(a, sh, set, it - 1);
                if (v == kAbort) return false;
            }
            RSTAMP(1, it, lane, wave);  // verdict of the previous iteration known
            if (v != kGoOn || it == a.budget) {
                const int done = v != kGoOn ? v : 0;
                finalize(set, s, done, true);
                if (blockIdx.x == 0 && wave == 0) {  // report: residual history (final since each barrier), outcome
                    const ResidentSync* sy = a.sync + set;
                    double* hist = a.b.res_hist + int64_t(set) * a.res_hist_stride;
                    int q0 = lane;
                    asm volatile("" : "+v"(q0));  // once per run: its addresses must not be carried (spilled) through the sweep loop
                    for (int q = q0; q < it; q += kWave)
                        if (a.sweep_begin + q < a.b.res_cap)
                            hist[a.sweep_begin + q] = residual_of(__hip_atomic_load(&sy->res[q], RLX_AGENT));
                    if (lane == 0) {
                        Ctl* hc = a.host_ctl + set;
                        hc->last_res = it > 0 ? residual_of(__hip_atomic_load(&sy->res[it - 1], RLX_AGENT)) : 0.0;
                        hc->n_sweeps = s;
                        hc->run_id = a.run_id;
                        hc->done = done;
                    }
                }
                active &= ~(1u << set);
                continue;
            }
            const double wres = phase(set, s);
            if (BATCH && a.direct == 0) arrive_batch(a, sh, set, it, s, wres, lane, wave);
            else arrive(a, sh, set, it, s, wres, lane, wave);
            RSTAMP(6, it, lane, wave);  // granules published
        }
        if (active == 0) break;
    }
    return true;
}

// A wave without a tile: takes part in the barriers, contributes nothing.
template <bool BATCH>
__device__ __forceinline__ bool resident_idle(const ResidentArgs& a, BlockShared& sh, int lane, int wave) {
    return resident_drive<BATCH>(a, sh, lane, wave, [](int, int) { return 0.0; }, [](int, int, int, bool) {});
}

// ---- the dataflow form: no grid barrier (single evidence set, more than one tile block) -----------------------
// Jacobi iteration i + 1 of a tile reads what its NEIGHBOUR tiles wrote in iteration i and overwrites what they read in
// iteration i (double buffers), nothing else (belief_propagation.hpp:78-101 read the old maps, :135-143 commit): a
// wave may start i + 1 as soon as every neighbour tile has finished i.  Each wave publishes, per iteration, a granule
// pair {generation | residual half} of its own; lane t of a waiting wave polls neighbour t's.  No __syncthreads, no
// block-level arrival: a slow wave holds up its neighbours only, and that skew averages out over the iterations
// instead of being paid at every barrier.
// The stop decision (:147) needs the maximum over ALL tiles, so it lags: the service block (here all its 8 waves)
// collects the granules of iteration i while the tiles compute i + 1 and publishes {generation, verdict}; a tile
// starts i + 2 only once the verdict of i is known.  When the verdict says "stop after i" a tile has at most
// executed i + 1 as well: that speculative iteration wrote the OTHER record / node buffer, so the state the run ends
// in is intact -- messages in the record buffer, pi(v) / lambda(v) in the node buffer every iteration stores them to
// (or still in registers when the tile had not started the speculative iteration).
// SHARD (a sharded engine, bn_create_sharded + bn_peer_import): the neighbour tiles across a cut edge live on other
// ranks -- other GPUs, or other processes / engines on this one.  Nothing changes in the protocol: a tile that touches
// a cut edge stores the message halves it produces for it into the peer's record buffer as well as its own (identical
// exchange-region layout on every rank, bn_plan.hpp) and its granules into the peer's table as well as its own; every
// rank's service block publishes the rank's residual to all ranks and takes the maximum.  Everything a rank READS is
// in its own memory.  Cross-rank traffic is system-scope (sc0 sc1) on fine-grained allocations.
#define RLX_SYS __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM
template <bool SHARD>
__device__ __forceinline__ unsigned long long flow_load(const unsigned long long* p) {
    if constexpr (SHARD) return __hip_atomic_load(p, RLX_SYS);
    else return __hip_atomic_load(p, RLX_AGENT);
}
template <bool SHARD>
__device__ __forceinline__ void flow_store(unsigned long long* p, unsigned long long v) {
    if constexpr (SHARD) __hip_atomic_store(p, v, RLX_SYS);
    else __hip_atomic_store(p, v, RLX_AGENT);
}
// granule pair of slot `slot` (= rank * kFlowSlotsPerRank + tile) for iterations of parity `par` in the table behind `f`
__device__ __forceinline__ unsigned long long* flow_slot(FlowSync* f, int nranks, int par, int slot) {
    return reinterpret_cast<unsigned long long*>(f + 1) + (size_t(par) * nranks * kFlowSlotsPerRank + slot) * 2;
}

// `where`: which wait gave up (diagnostics: the host prints it under BN_DEBUG) -- 1 a tile's neighbours / verdict, 2 the
// service block's own tiles, 3 the other ranks' residuals; bits 8.. the iteration
template <bool SHARD>
__device__ __forceinline__ void flow_raise_abort(const ResidentArgs& a, unsigned where = 0) {
    const unsigned long long w = (unsigned long long)(a.gen_base + 1u) | ((unsigned long long)kFlowAbort << 32);
    for (int q = 0; q < 8; ++q) flow_store<SHARD>(&a.flow->verdict[q].word, w);
    if constexpr (SHARD) {  // the other ranks stop at their next poll instead of running into their own deadline
        for (int r = 0; r < a.b.nranks; ++r)
            if (r != a.b.rank)
                for (int q = 0; q < 8; ++q) flow_store<true>(&a.peers[r].flow->verdict[q].word, w);
    }
    __hip_atomic_store(a.host_abort, 0x80000000u | where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Waits until this tile may run iteration `it`; returns the verdict that ends the run for it (kFlowGoOn: run the
// iteration) and, with a stop verdict, the number of iterations the run consists of in n_it.
template <bool SHARD>
__device__ __forceinline__ unsigned flow_wait(const ResidentArgs& a, int tile, int it, int lane, int& n_it) {
    n_it = 0;
    if (it == 0) return kFlowGoOn;
    FlowSync* f = a.flow;
    // lane t polls neighbour t (+ 64, + 128, ... for the few tiles with more than 64 neighbours: a grid's first column);
    // a round whose neighbours were all seen ready is not polled again
    const int32_t* nbr_list = a.nbr + int64_t(tile) * a.nbr_chunks * kWave + lane;
    int round = 0;
    int nbr = nbr_list[0];
    const unsigned long long* vword = &f->verdict[blockIdx.x & 7].word;
    // A neighbour's slot of this parity carries EXACTLY the generation of iteration it - 1 once it has finished it (it
    // cannot finish it + 1 before this tile has finished `it`); ranks whose generation counters have drifted apart
    // (one of them restarted after an aborted launch) never match and run into the deadline instead of reading on.
    const unsigned want_nb = a.gen_base + unsigned(it);
    const unsigned want_v = a.gen_base + unsigned(it) - 1u;  // iteration it - 2 is decided
    // the 100 MHz clock is a scalar MEMORY read (s_memrealtime, a round trip of its own): it bounds the wait, so it is
    // looked at on every 32nd unsuccessful poll only
    unsigned long long t0 = 0;
    for (unsigned polls = 0;; ++polls) {
        bool nb_ok = true;
        if (nbr >= 0) nb_ok = unsigned(flow_load<SHARD>(flow_slot(f, a.b.nranks, (it - 1) & 1, nbr)) >> 32) == want_nb;
        if (__all(nb_ok) && round + 1 < a.nbr_chunks) {  // wave-uniform: on to the next 64 neighbours
            ++round;
            nbr = nbr_list[round * kWave];
            if (__any(nbr >= 0)) continue;
        }
        const unsigned long long w = flow_load<SHARD>(vword);
        const unsigned gen = unsigned(w), kind = unsigned(w >> 32);
        const bool ours = gen - (a.gen_base + 1u) < unsigned(kResidentBudget);  // published by THIS launch (generations count on across launches)
        if (ours && kind != kFlowGoOn) {
            n_it = int(gen - a.gen_base);
            return kind;
        }
        const bool v_ok = it < 2 || (ours && gen >= want_v);
        if (it < a.budget && v_ok && __all(nb_ok)) return kFlowGoOn;
        if ((polls & 31u) == 31u) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > a.timeout_ticks) {
                if (lane == 0) flow_raise_abort<SHARD>(a, 1u | (unsigned(it) << 8) | (unsigned(tile) << 20));
                return kFlowAbort;
            }
        }
        // back off between polls: every poll is a handful of L2-missing line requests that queue with the message
        // traffic of the waves still computing (measured: polling flat out slows their roles down by ~1 us)
        for (int z = 0; z < a.poll_sleep; ++z) __builtin_amdgcn_s_sleep(8);
    }
}

template <bool SHARD, class Phase, class Finalize>
__device__ __forceinline__ bool flow_drive(const ResidentArgs& a, int tile, int lane, int wave, Phase&& phase, Finalize&& finalize) {
    const int my_slot = a.b.rank * kFlowSlotsPerRank + tile;
    unsigned pub = 0;  // ranks that hold a neighbour of this tile
    if constexpr (SHARD) pub = a.pub_mask[tile];
    for (int it = 0;; ++it) {
        RSTAMP(0, it, lane, wave);
        int n_it;
        const unsigned v = flow_wait<SHARD>(a, tile, it, lane, n_it);
        RSTAMP(1, it, lane, wave);  // neighbours ready, verdict of it - 2 known
        if (v == kFlowAbort) return false;
        if (v != kFlowGoOn) {  // the run consists of n_it iterations of this launch; this wave has executed `it` (n_it or n_it + 1)
            finalize(0, a.sweep_begin + n_it, v == kFlowBudget ? 0 : int(v), n_it == it);
            return true;
        }
        const double wres = phase(0, a.sweep_begin + it);
        const unsigned long long bits = wave_umax((unsigned long long)__double_as_longlong(wres));
        RSTAMP(3, it, lane, wave);  // sweep issued
        drain_stores();             // this wave's write-through stores -- local and peer -- have reached memory
        RSTAMP(4, it, lane, wave);
        const unsigned gen = a.gen_base + unsigned(it) + 1u;
        if (lane == 0) {
            unsigned long long* g = flow_slot(a.flow, a.b.nranks, it & 1, my_slot);
            flow_store<SHARD>(g, granule(gen, unsigned(bits >> 32)));
            flow_store<SHARD>(g + 1, granule(gen, unsigned(bits)));
        }
        if constexpr (SHARD) {  // lane q + 1 tells rank q (only the generation matters there: the residual travels by rank)
            const int q = lane - 1;
            if (q >= 0 && q < a.b.nranks && ((pub >> q) & 1u) != 0) {
                unsigned long long* g = flow_slot(a.peers[q].flow, a.b.nranks, it & 1, my_slot);
                flow_store<true>(g, granule(gen, 0u));
            }
        }
        RSTAMP(6, it, lane, wave);  // granules published
    }
}

// The service block of the dataflow form, all kResidentWaves waves: thread x sweeps the granule pairs of tiles x, x + 512,
// ... of the iteration in hand until they carry its generation; the block reduces, thread 0 decides and publishes.
// SHARD: thread 0 first publishes this rank's maximum to every rank and waits for theirs; all ranks decide alike.
template <bool SHARD>
__device__ __forceinline__ void flow_service(const ResidentArgs& a, BlockShared& sh, int lane, int wave) {
    FlowSync* f = a.flow;
    const int nt = a.b.n_tiles, nranks = a.b.nranks, base = a.b.rank * kFlowSlotsPerRank;
    const unsigned long long t_first = wall_clock64();
    if (threadIdx.x == 0) sh.verdict[1] = 0;
    __syncthreads();
    int n_it = 0;
    unsigned v = kFlowGoOn;
    for (int it = 0; it < a.budget; ++it) {
        const unsigned gen = a.gen_base + unsigned(it) + 1u;
        unsigned long long m = 0;
        unsigned long long t0 = 0;
        bool ok = true;
        for (unsigned polls = 0;; ++polls) {
            bool mine = true;
            unsigned long long acc = 0;
            for (int t = threadIdx.x; t < nt; t += a.waves * kWave) {
                const unsigned long long* g = flow_slot(f, nranks, it & 1, base + t);
                const unsigned long long hi = flow_load<SHARD>(g);
                const unsigned long long lo = flow_load<SHARD>(g + 1);
                mine = mine && unsigned(hi >> 32) == gen && unsigned(lo >> 32) == gen;
                const unsigned long long x = (hi << 32) | (lo & 0xffffffffull);
                acc = x > acc ? x : acc;
            }
            m = acc;
            if (__all(mine)) break;
            const unsigned long long w = flow_load<SHARD>(&f->verdict[0].word);
            if (unsigned(w >> 32) == kFlowAbort && unsigned(w) - (a.gen_base + 1u) < unsigned(kResidentBudget)) { ok = false; break; }
            if ((polls & 31u) == 31u) {
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > a.timeout_ticks) {
                    if (lane == 0) flow_raise_abort<SHARD>(a, 2u | (unsigned(it) << 8));
                    ok = false;
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(1);
        }
        m = wave_umax(m);
        if (lane == 0) {
            sh.slot[0][wave] = m;
            if (!ok) sh.verdict[1] = 1;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned decision = kFlowAbort;
            bool good = sh.verdict[1] == 0;
            unsigned long long mm = 0;
            if (good) {
                for (int w = 0; w < a.waves; ++w) mm = sh.slot[0][w] > mm ? sh.slot[0][w] : mm;
            }
            if constexpr (SHARD) {
                if (good) {  // this rank's maximum to every rank (its own table included), then the maximum over all of them
                    for (int r = 0; r < nranks; ++r) {
                        unsigned long long* g = &a.peers[r].flow->rank_granule[it & 1][a.b.rank][0];
                        flow_store<true>(g, granule(gen, unsigned(mm >> 32)));
                        flow_store<true>(g + 1, granule(gen, unsigned(mm)));
                    }
                    unsigned long long t1 = 0;
                    for (unsigned polls = 0; good; ++polls) {
                        bool all = true;
                        unsigned long long acc = 0;
                        for (int r = 0; r < nranks; ++r) {
                            const unsigned long long hi = flow_load<true>(&f->rank_granule[it & 1][r][0]);
                            const unsigned long long lo = flow_load<true>(&f->rank_granule[it & 1][r][1]);
                            all = all && unsigned(hi >> 32) == gen && unsigned(lo >> 32) == gen;
                            const unsigned long long x = (hi << 32) | (lo & 0xffffffffull);
                            acc = x > acc ? x : acc;
                        }
                        if (all) { mm = acc; break; }
                        const unsigned long long w = flow_load<true>(&f->verdict[0].word);
                        if (unsigned(w >> 32) == kFlowAbort && unsigned(w) - (a.gen_base + 1u) < unsigned(kResidentBudget)) good = false;
                        if ((polls & 31u) == 31u) {
                            const unsigned long long now = wall_clock64();
                            if (t1 == 0) t1 = now;
                            if (now - t1 > a.timeout_ticks) {
                                flow_raise_abort<true>(a, 3u | (unsigned(it) << 8));
                                good = false;
                            }
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            if (good) {
                decision = unsigned(verdict_of(a, residual_of(mm), a.sweep_begin + it + 1));
                if (decision == kFlowGoOn && it == a.budget - 1) decision = kFlowBudget;
                flow_store<SHARD>(&f->res[it], mm);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the residual is recorded before anyone learns the verdict
                const unsigned long long word = (unsigned long long)gen | ((unsigned long long)decision << 32);
                for (int q = 0; q < 8; ++q) flow_store<SHARD>(&f->verdict[q].word, word);
            }
            sh.verdict[0] = int(decision);
        }
        __syncthreads();
        v = unsigned(sh.verdict[0]);
        n_it = it + 1;
        if (v != kFlowGoOn) break;
    }
    if (wave != 0) return;
    // report: residual history, outcome, device clock
    if (v != kFlowAbort) {
        double* hist = a.b.res_hist;
        for (int q = lane; q < n_it; q += kWave)
            if (a.sweep_begin + q < a.b.res_cap) hist[a.sweep_begin + q] = residual_of(flow_load<SHARD>(&f->res[q]));
    }
    if (lane == 0) {
        Ctl* hc = a.host_ctl;
        hc->last_res = (v != kFlowAbort && n_it > 0) ? residual_of(flow_load<SHARD>(&f->res[n_it - 1])) : 0.0;
        hc->n_sweeps = a.sweep_begin + n_it;
        hc->t_first = t_first;
        hc->t_last = wall_clock64();
        hc->run_id = a.run_id;
        hc->done = v == kFlowAbort ? -1 : (v == kFlowBudget ? 0 : int(v));
    }
}

// Tables of more than 32 entries (k = 4 with two parents: 64) keep the entries of the upper half of the
// own states in LDS -- 16 bytes per lane and slot, lane-contiguous: conflict-free ds_read_b128 -- and the
// lower half in registers; 128 VGPRs of CPT plus the working set do not fit 256 registers, and the
// compiler's answer, scratch memory, would re-read two thirds of the table through the caches each sweep.
// BIG: the block runs at most four waves, one per SIMD, and its kernel may use the whole register file: the CPT stays in
// registers entirely (no LDS slots) and the contraction is left to the scheduler (no pins) -- a wave alone on its SIMD
// has nobody to fill its dependency stalls, so instruction-level parallelism is what it runs on.
template <int K, int M, int RC, bool BATCH, bool FLOW, bool SHARD = false, bool BIG = false>
__device__ __forceinline__ bool resident_tile(const ResidentArgs& a, BlockShared& sh, const TileDesc& td, int tile, int lane, int wave,
                                              double2_t* cpt_lds) {
    static_assert(!(BATCH && FLOW), "the dataflow form runs one evidence set");
    static_assert(!SHARD || FLOW, "shards exchange through the dataflow form");
    constexpr int KP = (K + 1) & ~1, H = KP / 2;
    constexpr int C = ipow(K, M), S = K * C, SP = (S + 1) & ~1;
    constexpr int CB = (M > 0) ? C / K : 0;
    // entries [0, SR) of the i-major image stay in registers, the rest in LDS: 28 of 64 (k = 4, two parents) leaves the
    // working set just enough registers; everything for smaller tables
    constexpr int SR = (S > 32 && !BIG) ? 28 : SP;
    static_assert(SR % 2 == 0 && (SR == SP || (S - SR) / 2 <= kResidentLdsSlots), "CPT split");
    auto PIN = [](double& x) { if constexpr (!BIG) pin_here(x); };
    const BpBuffers& b = a.b;
    const bool active = lane < td.n_nodes;
    const int lc = active ? lane : 0;  // idle lanes shadow lane 0 and store nothing

    // ---- resident state: CPT, evidence mark, references, pi(v), lambda(v)
    const double2_t* cp = reinterpret_cast<const double2_t*>(b.cpt + td.cpt_base) + lc;
    double cpt[SR];
#pragma unroll
    for (int q = 0; q < SR / 2; ++q) {
        const double2_t x = cp[q * kWave];
        cpt[2 * q] = x.x;
        cpt[2 * q + 1] = x.y;
    }
    if constexpr (SR < SP) {
#pragma unroll
        for (int q = SR / 2; q < S / 2; ++q) cpt_lds[(q - SR / 2) * kWave + lane] = cp[q * kWave];
    }
    bool frozen = false;  // evidence mark of this lane's node for the set in hand
    const int64_t rbase = td.rec_base / 2 + lc;  // this lane's slot in the tile's record block (double2 units)
    const MsgRef* orf = b.out_refs + td.out_base + lc;
    // SHARD: a tile behind the interior ones touches a cut edge (wave-uniform).  Its in-edges are reached through
    // references (a parent on another rank: both halves in the exchange region, contiguous chunks), the others sit in
    // the tile's own record block as everywhere else.
    const bool btile = SHARD && tile >= a.n_interior;
    const bool in_by_ref = SHARD && td.in_ref_base >= 0;
    // (fetched at each use -- three times a sweep, boundary tiles only -- rather than kept: registers matter here)
    auto IREF = [&](int j) -> MsgRef { return b.in_refs[td.in_ref_base + j * kWave + lc]; };
    // chunk h of the pi-message (part 0) / lambda-message (part 1) of in-edge j, double2 units
    auto in_idx = [&](int j, int part, int h) -> int64_t {
        if constexpr (SHARD) {
            if (in_by_ref) {
                const MsgRef r = IREF(j);
                const bool cut = r.lam < 0;
                const int lam = cut ? ~r.lam : r.lam;
                const int stride = cut ? 1 : (lam - r.pi) / H;
                return int64_t((part ? lam : r.pi) + h * stride);
            }
        }
        return rbase + ((j * 2 + part) * H + h) * kWave;
    };
    // record access of this tile: system scope where a peer may be the other end
    auto LD = [&](__amdgpu_buffer_rsrc_t r, int64_t idx2) -> double2_t {
        if constexpr (SHARD) {
            if (btile) return ld_rec_sys(r, idx2);
        }
        return ld_rec(r, idx2);
    };
    auto ST = [&](__amdgpu_buffer_rsrc_t r, int64_t idx2, double2_t v) {
        if constexpr (SHARD) {
            if (btile) { st_rec_sys(r, idx2, v); return; }
        }
        st_rec(r, idx2, v);
    };
    // A message half this rank produced for a cut edge goes into the peer's exchange region too (same offset inside the
    // region: its layout is identical on every rank).  `other` = where the edge's other half lives: the segment of the rank across the cut.
    // One pass per rank that some lane of the wave has to reach (a stripe boundary: one).
    auto push_remote = [&](bool cut_lane, int64_t other, bool into_rec1, const int64_t (&idx)[H], const double2_t (&val)[H]) {
        if constexpr (SHARD) {
            if (!btile) return;
            for (int q = 0; q < b.nranks; ++q) {
                if (q == b.rank) continue;
                const int64_t lo = b.g_base + int64_t(q) * b.seg_d2;
                const bool mine = cut_lane && other >= lo && other < lo + b.seg_d2;
                if (__any(mine)) {
                    double* base = into_rec1 ? a.peers[q].rec1 : a.peers[q].rec0;
                    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(base, 0, int(a.peers[q].rec_bytes), 0x00020000);
                    const int64_t shift = a.peers[q].g_base - b.g_base;  // same place in THEIR exchange region
                    if (mine) {
#pragma unroll
                        for (int h = 0; h < H; ++h) st_rec_sys(pr, idx[h] + shift, val[h]);
                    }
                }
            }
        }
    };
    // Out-edge references, packed (8 bytes per child) and decoded at each use: registers matter here.
    // Up to 4 children they stay resident; beyond (RC = 8) they are fetched again every sweep -- one
    // more dependent load at the head of those tiles' sweep, 16 registers less across the whole loop.
    constexpr bool kRefsResident = RC <= 4;
    MsgRef oref[RC > 0 ? RC : 1];
#pragma unroll
    for (int c = 0; c < RC; ++c) {
        oref[c] = MsgRef{-1, 0};
        if (kRefsResident && active && c < td.cmax) oref[c] = orf[c * kWave];
    }
    // pi(v), lambda(v).  One set per launch: they live in these registers from the first sweep to the last.
    // Several sets: a set's vectors are read from its node buffer at the start of its turn and written back
    // at the end (buffer parity as on the launch path), the registers are only the turn's working copy.
    double piv[KP], lav[KP];
    auto load_nodes = [&](int set, int s) {
        frozen = b.frozen[(BATCH ? int64_t(set) * a.slot_stride : 0) + td.slot_base + lc] == b.frozen_mark;
#pragma unroll
        for (int i = 0; i < KP; ++i) {  // initial state (:38-64): roots start from their CPT row
            piv[i] = (M == 0 && i < K) ? cpt[i] : 1.0;
            lav[i] = 1.0;
        }
        // evidence nodes hold their vector as pi and lambda (:68-73); from the second sweep on (and in a
        // continued run) the vectors are what the previous sweep left
        if (frozen || s > 0) {
            const double2_t* nin = reinterpret_cast<const double2_t*>(((s & 1) ? b.node1 : b.node0) + (BATCH ? int64_t(set) * a.node_stride : 0) +
                                                                      td.node_base) + lc;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const double2_t x = nin[h * kWave], y = nin[(H + h) * kWave];
                piv[2 * h] = x.x; piv[2 * h + 1] = x.y;
                lav[2 * h] = y.x; lav[2 * h + 1] = y.y;
            }
        }
    };
    auto store_nodes = [&](int set, int n) {  // into the buffer the reader of state n expects
        if (!active) return;
        double2_t* nout = reinterpret_cast<double2_t*>(((n & 1) ? b.node1 : b.node0) + (BATCH ? int64_t(set) * a.node_stride : 0) + td.node_base) + lane;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            double2_t y, z;
            y.x = piv[2 * h]; y.y = (2 * h + 1 < K) ? piv[2 * h + 1] : 0.0;
            z.x = lav[2 * h]; z.y = (2 * h + 1 < K) ? lav[2 * h + 1] : 0.0;
            nout[h * kWave] = y;
            nout[(H + h) * kWave] = z;
        }
    };
    if constexpr (!BATCH) load_nodes(0, a.sweep_begin);

    const size_t rec_bytes = size_t(b.rec_total_doubles) * 8;
    auto phase = [&](int set, int s) -> double {
        if constexpr (BATCH) load_nodes(set, s);
        const bool first = s == 0;
        const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(b.rec0 + (BATCH ? int64_t(set) * a.rec_stride : 0), 0, int(rec_bytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(b.rec1 + (BATCH ? int64_t(set) * a.rec_stride : 0), 0, int(rec_bytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t rin = (s & 1) ? rsrc1 : rsrc0;
        const __amdgpu_buffer_rsrc_t rout = (s & 1) ? rsrc0 : rsrc1;

        double wres = 0.0;
        // ---- parent role first (it needs only the OLD pi(v) and the children's lambda-messages; done
        // before the child role so that its registers are free again when the CPT products start):
        // lambda(v) (:220-238) and the pi-message to every child (:202-218) = pi(v) times the OTHER
        // children's lambda-messages, ascending child order
        double lan[K];
        {
            if constexpr (!kRefsResident) {
#pragma unroll
                for (int c = 0; c < RC; ++c) {
                    oref[c] = MsgRef{-1, 0};
                    if (active && c < td.cmax) oref[c] = orf[c * kWave];
                }
            }
            double lkc[RC > 0 ? RC : 1][KP];
#pragma unroll
            for (int c = 0; c < RC; ++c)
#pragma unroll
                for (int i = 0; i < KP; ++i) lkc[c][i] = 1.0;
            if (!first) {
#pragma unroll
                for (int c = 0; c < RC; ++c)
                    if (c < td.cmax) {
                        const Loc l = decode_ref(oref[c], H);  // a missing child reads record 0 and contributes 1.0
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            const double2_t y = LD(rin, l.lam + h * l.stride);
                            lkc[c][2 * h] = l.has ? y.x : 1.0;
                            lkc[c][2 * h + 1] = l.has ? y.y : 1.0;
                        }
                    }
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                double acc = 1.0;
#pragma unroll
                for (int c = 0; c < RC; ++c) acc *= lkc[c][i];
                lan[i] = acc;
            }
            normalize_k<K>(lan);
#pragma unroll
            for (int c = 0; c < RC; ++c) {
                if (c < td.cmax) {  // wave-uniform
                    PIN(wres);  // one child at a time (see pin_here)
#pragma unroll
                    for (int i = 0; i < K; ++i) PIN(piv[i]);

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
                    const Loc l = decode_ref(oref[c], H);
                    if (l.has) {
                        double old[KP], o[KP];
#pragma unroll
                        for (int i = 0; i < KP; ++i) { old[i] = 1.0; o[i] = 0.0; }
                        if (!first) {
#pragma unroll
                            for (int h = 0; h < H; ++h) {  // previous pi-message of this edge, for the residual
                                const double2_t x = LD(rin, l.pi + h * l.stride);
                                old[2 * h] = x.x; old[2 * h + 1] = x.y;
                            }
                        }
#pragma unroll
                        for (int i = 0; i < K; ++i) {
                            o[i] = u[i];
                            wres = res_acc(wres, fabs(u[i] - old[i]));
                        }
                        int64_t at[H];
                        double2_t yv[H];
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            yv[h].x = o[2 * h]; yv[h].y = o[2 * h + 1];
                            at[h] = l.pi + h * l.stride;
                            ST(rout, at[h], yv[h]);
                        }
                        if constexpr (SHARD) push_remote(oref[c].lam < 0, l.lam, (s & 1) == 0, at, yv);  // the child's owner reads this half
                    }
                }
            }
        }

        RSTAMP(7, s - a.sweep_begin, lane, wave);  // parent role
        __builtin_amdgcn_sched_barrier(0);
        // ---- child role, calculate_pi (:174-200) and calculate_lambda_k (:240-266): each accumulator
        // sees its terms in the reference's order (see tile_uniform); lambda(v)[i] * cpt is formed at each
        // use instead of being kept (same operation, same bits, 2 C fewer live registers)
        double pim[M > 0 ? M : 1][KP];
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = 0; i < KP; ++i) pim[j][i] = 1.0;
        if (!first) {
#pragma unroll
            for (int j = 0; j < M; ++j)
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const double2_t x = LD(rin, in_idx(j, 0, h));
                    pim[j][2 * h] = x.x; pim[j][2 * h + 1] = x.y;
                }
        }
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
                // one own state at a time: everything this step consumes is pinned here
                PIN(lav[ib]);
#pragma unroll
                for (int j = 0; j < M; ++j)
#pragma unroll
                    for (int i = 0; i < K; ++i) PIN(pim[j][i]);
#pragma unroll
                for (int jt = 0; jt < M; ++jt)
#pragma unroll
                    for (int ct = 0; ct < K; ++ct) PIN(out[jt][ct]);
                // entry `cond` of own state ib: a register, or this wave's LDS slots (read where it is used:
                // a copy of the whole row would cost 2 C registers for the length of the step)
                const double* lds_lane = reinterpret_cast<const double*>(cpt_lds + lane);
                auto ROW = [&](int cond) -> double {
                    const int e = ib * C + cond;  // compile-time after unrolling
                    if (e < SR) return cpt[e < SR ? e : 0];
                    return lds_lane[((e - SR) >> 1) * (2 * kWave) + (e & 1)];
                };
                double acc = 0.0;
#pragma unroll
                for (int rr = 0; rr < CB; ++rr) {
                    if (rr > 0) {  // K + M K independent chains per step are plenty; pin the step's boundary
                        if constexpr (!BIG) asm volatile("" ::: "memory");
                        PIN(acc);
                        PIN(lav[ib]);
#pragma unroll
                        for (int j = 0; j < M; ++j)
#pragma unroll
                            for (int i = 0; i < K; ++i) PIN(pim[j][i]);
#pragma unroll
                        for (int jt = 0; jt < M; ++jt)
#pragma unroll
                            for (int ct = 0; ct < K; ++ct) PIN(out[jt][ct]);
                    }
#pragma unroll
                    for (int x = 0; x < K; ++x) {
                        const int cond = rr * K + x;
                        double value = ROW(cond);
#pragma unroll
                        for (int j = 0; j < M; ++j) value *= pim[j][(cond / ipow(K, M - 1 - j)) % K];
                        acc += value;
                    }
#pragma unroll
                    for (int jt = 0; jt < M; ++jt) {
                        const int stride = ipow(K, M - 1 - jt);
#pragma unroll
                        for (int ct = 0; ct < K; ++ct) {
                            const int cond = (rr / stride) * stride * K + ct * stride + (rr % stride);
                            double value = lav[ib] * ROW(cond);
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
        RSTAMP(8, s - a.sweep_begin, lane, wave);  // contraction
        __builtin_amdgcn_sched_barrier(0);
        normalize_k<K>(pin);
#pragma unroll
        for (int jt = 0; jt < M; ++jt) normalize_k<K>(out[jt]);

        // ---- lambda-messages out + residual (:105-131); this lane's previous messages are re-read from the old buffer
#pragma unroll
        for (int jt = 0; jt < M; ++jt) {
            double old[KP], o[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i) { old[i] = 1.0; o[i] = 0.0; }
            if (!first) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const double2_t y = LD(rin, in_idx(jt, 1, h));
                    old[2 * h] = y.x; old[2 * h + 1] = y.y;
                }
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                o[i] = out[jt][i];
                wres = res_acc(wres, fabs(out[jt][i] - old[i]));
            }
            if (active) {
                int64_t at[H];
                double2_t yv[H];
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    yv[h].x = o[2 * h]; yv[h].y = o[2 * h + 1];
                    at[h] = in_idx(jt, 1, h);
                    ST(rout, at[h], yv[h]);
                }
                if constexpr (SHARD) {
                    if (in_by_ref) {  // the parent's owner reads this half
                        const MsgRef r = IREF(jt);
                        push_remote(r.lam < 0, r.pi, (s & 1) == 0, at, yv);
                    }
                }
            }
        }
        if (!active) wres = 0.0;
        // the node's new vectors; evidence nodes keep theirs (:177, :223)
        if (!frozen) {
#pragma unroll
            for (int i = 0; i < K; ++i) { piv[i] = pin[i]; lav[i] = lan[i]; }
        }
        // several sets: a set's vectors go through memory between its turns; dataflow form: the state the run stops in
        // may be one iteration older than the registers (the stop decision lags)
        if constexpr (BATCH || FLOW) store_nodes(set, s + 1);
        return wres;
    };
    // the set stopped after n sweeps (or the budget ran out, done == 0): node vectors to the buffer the next
    // reader expects (parity of n), beliefs = normalize(pi % lambda) (:151-158) when the run is over
    auto finalize = [&](int set, int n, int done, bool regs_hold_n) {
        if constexpr (BATCH) load_nodes(set, n);
        else if constexpr (FLOW) { if (!regs_hold_n) load_nodes(set, n); }  // every iteration stored its vectors
        else store_nodes(set, n);
        if (active && done != 0) {
            const int64_t boff = b.slot_boff[td.slot_base + lane];
            double* beliefs = b.beliefs + (BATCH ? int64_t(set) * a.belief_stride : 0);
            double bel[K];
            double sum = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                bel[i] = piv[i] * lav[i];
                sum += bel[i];
            }
#pragma unroll
            for (int i = 0; i < K; ++i) beliefs[boff + i] = bel[i] / sum;
        }
    };
    if constexpr (FLOW) return flow_drive<SHARD>(a, tile, lane, wave, phase, finalize);
    else return resident_drive<BATCH>(a, sh, lane, wave, phase, finalize);
}

// MODE: 0 = one evidence set, grid barrier per sweep (one-block grids: LDS only); 1 = several evidence sets per launch
// (node vectors through memory between a set's turns); 2 = one evidence set, dataflow form (no grid barrier)
enum : int { kModeBarrier = 0, kModeBatch = 1, kModeFlow = 2, kModeFlowShard = 3 };  // 3: dataflow form of a sharded engine (peer stores)

template <int K, int M, int MODE, int LEAN, bool BIG>
__device__ __forceinline__ bool resident_dispatch(const ResidentArgs& a, BlockShared& sh, const TileDesc& td, int tile, int lane, int wave,
                                                  double2_t* cpt_lds) {
    constexpr bool BATCH = MODE == kModeBatch, FLOW = MODE == kModeFlow || MODE == kModeFlowShard, SHARD = MODE == kModeFlowShard;
#ifdef BN_RES_ONLY_RC  // experiments: one instantiation only
    return resident_tile<K, M, BN_RES_ONLY_RC, BATCH, FLOW, SHARD, BIG>(a, sh, td, tile, lane, wave, cpt_lds);
#else
    if constexpr (LEAN != 0) return resident_tile<K, M, 2, BATCH, FLOW, SHARD, BIG>(a, sh, td, tile, lane, wave, cpt_lds);
    if (td.cmax <= 2) return resident_tile<K, M, 2, BATCH, FLOW, SHARD, BIG>(a, sh, td, tile, lane, wave, cpt_lds);
    if (td.cmax <= 4) return resident_tile<K, M, 4, BATCH, FLOW, SHARD, BIG>(a, sh, td, tile, lane, wave, cpt_lds);
    return resident_tile<K, M, 8, BATCH, FLOW, SHARD, BIG>(a, sh, td, tile, lane, wave, cpt_lds);  // the host admits <= 8 children per node
#endif
}

// LEAN = k in {2, 3, 4}: every node has arity k and at most 2 children (grids, chains, polytrees of that
// shape: the headline workload) -- an instantiation that carries no code or registers for the other shapes,
// so its code-object figures (0 spills, tests/test_host_logic.py) are those of the path that actually runs.
// LEAN = 0: every shape the resident path admits.
// WMAX: most waves a block of this instantiation is launched with.  8 (two waves per SIMD: 256 registers each) everywhere
// but in the all-shapes instantiations for networks that run four waves per block: one wave per SIMD may use the whole
// register file (512), which is what the 4- and 8-children parent roles inlined side by side need to stay out of scratch.
template <int MODE, int LEAN, int WMAX = kResidentWaves>
__global__ __launch_bounds__(WMAX * kWave, WMAX == kResidentWaves ? 2 : 1) void bp_resident_kernel(ResidentArgs a) {
    constexpr bool BATCH = MODE == kModeBatch, FLOW = MODE == kModeFlow || MODE == kModeFlowShard, SHARD = MODE == kModeFlowShard;
    __shared__ BlockShared sh;
    __shared__ double2_t cpt_lds_all[WMAX][kResidentLdsSlots * kWave];  // the CPT entries not kept in registers, 18 KiB per wave
    const BpBuffers& b = a.b;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) { sh.t_arrive = 0; sh.skew_ticks = 0; }
    if constexpr (BATCH) {
        if (threadIdx.x < kResidentMaxSets) { sh.arrived[threadIdx.x] = 0; sh.poll_it[threadIdx.x] = -1; sh.ver[threadIdx.x] = -1; }
        __syncthreads();
    }
    if constexpr (!FLOW) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {  // written now rather than kept in registers for the whole run
            const unsigned long long t_first = wall_clock64();
            for (int set = 0; set < a.n_sets; ++set) a.host_ctl[set].t_first = t_first;
        }
    }
    // blocks past the tile blocks serve the barrier / collect the residuals
    const int nb = a.n_tile_blocks;
    if (int(blockIdx.x) >= nb) {
        if constexpr (FLOW) flow_service<SHARD>(a, sh, lane, wave);
        else if (wave == 0) resident_service(a, lane);
        return;
    }
    // XCD-contiguous tile mapping (speed only), as in the per-sweep kernel
    const int lb = (nb % 8 == 0) ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int tile = lb * a.waves + wave;  // a.waves = 8, or 4 when that still gives every tile a wave: one wave per SIMD then
    bool ok;
    if (tile >= b.n_tiles) {
        if constexpr (FLOW) return;  // nobody waits for a wave without a tile
        else ok = resident_idle<BATCH>(a, sh, lane, wave);
    } else {
        const TileDesc td = b.tiles[tile];
        double2_t* lds = cpt_lds_all[wave];
        {
#ifdef BN_RES_ONLY_K
            ok = resident_dispatch<BN_RES_ONLY_K, BN_RES_ONLY_M, MODE, LEAN, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds);
#else
            if constexpr (LEAN != 0) {
                switch (td.m) {
                    case 0: ok = resident_dispatch<LEAN, 0, MODE, LEAN, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds); break;
                    case 1: ok = resident_dispatch<LEAN, 1, MODE, LEAN, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds); break;
                    default: ok = resident_dispatch<LEAN, 2, MODE, LEAN, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds); break;
                }
            } else {
                switch (td.kv * 8 + td.m) {
                    case 2 * 8 + 0: ok = resident_dispatch<2, 0, MODE, 0, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds); break;
                    case 2 * 8 + 1: ok = resident_dispatch<2, 1, MODE, 0, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds); break;
                    case 2 * 8 + 2: ok = resident_dispatch<2, 2, MODE, 0, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds); break;
                    case 3 * 8 + 0: ok = resident_dispatch<3, 0, MODE, 0, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds); break;
                    case 3 * 8 + 1: ok = resident_dispatch<3, 1, MODE, 0, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds); break;
                    case 3 * 8 + 2: ok = resident_dispatch<3, 2, MODE, 0, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds); break;
                    case 4 * 8 + 0: ok = resident_dispatch<4, 0, MODE, 0, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds); break;
                    case 4 * 8 + 1: ok = resident_dispatch<4, 1, MODE, 0, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds); break;
                    default: ok = resident_dispatch<4, 2, MODE, 0, (WMAX < kResidentWaves)>(a, sh, td, tile, lane, wave, lds); break;  // host admits only the shapes above
                }
            }
#endif
        }
    }
    if constexpr (!FLOW) {
        // a block that gave up a bounded wait says so itself: block 0 may long have reported its own outcome
        if (!ok && threadIdx.x == 0 && a.host_abort) __hip_atomic_store(a.host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            // per-set outcomes were written as the sets stopped; the launch as a whole: timing, and the abort mark
            const unsigned long long t_last = wall_clock64();
            for (int set = 0; set < a.n_sets; ++set) {
                a.host_ctl[set].t_last = t_last;
                if (!ok) { a.host_ctl[set].run_id = a.run_id; a.host_ctl[set].done = -1; }
            }
        }
    }
}

// lean_k: 2, 3 or 4 when every node has that arity and at most 2 children, else 0
int launch_bp_resident(const ResidentArgs& a, int grid_blocks, int lean_k, void* stream) {
    (void)hipGetLastError();  // drop any stale error of this thread
    const dim3 g(grid_blocks), t(a.waves * kWave);
    hipStream_t s = (hipStream_t)stream;
    const int mode = a.n_sets > 1 ? kModeBatch : (a.flow != nullptr ? (a.peers != nullptr ? kModeFlowShard : kModeFlow) : kModeBarrier);
#define BN_RES_LAUNCH(L)                                                                          \
    switch (mode) {                                                                               \
        case kModeBatch: hipLaunchKernelGGL((bp_resident_kernel<kModeBatch, L>), g, t, 0, s, a); break;   \
        case kModeFlow: hipLaunchKernelGGL((bp_resident_kernel<kModeFlow, L>), g, t, 0, s, a); break;     \
        case kModeFlowShard: hipLaunchKernelGGL((bp_resident_kernel<kModeFlowShard, L>), g, t, 0, s, a); break; \
        default: hipLaunchKernelGGL((bp_resident_kernel<kModeBarrier, L>), g, t, 0, s, a); break;         \
    }
#define BN_RES_LAUNCH_W4(L)                                                                       \
    switch (mode) {                                                                               \
        case kModeBatch: hipLaunchKernelGGL((bp_resident_kernel<kModeBatch, L, 4>), g, t, 0, s, a); break;   \
        case kModeFlow: hipLaunchKernelGGL((bp_resident_kernel<kModeFlow, L, 4>), g, t, 0, s, a); break;     \
        case kModeFlowShard: hipLaunchKernelGGL((bp_resident_kernel<kModeFlowShard, L, 4>), g, t, 0, s, a); break; \
        default: hipLaunchKernelGGL((bp_resident_kernel<kModeBarrier, L, 4>), g, t, 0, s, a); break;         \
    }
    const bool w4 = a.waves <= 4 && grid_blocks > 1;
    switch (lean_k) {
        case 2: BN_RES_LAUNCH(2); break;  // (k = 2, 3: the tables are in registers anyway and nothing is pinned hard)
        case 3: BN_RES_LAUNCH(3); break;
        case 4: if (w4) { BN_RES_LAUNCH_W4(4); } else { BN_RES_LAUNCH(4); } break;
        default: if (w4) { BN_RES_LAUNCH_W4(0); } else { BN_RES_LAUNCH(0); } break;
    }
#undef BN_RES_LAUNCH_W4
#undef BN_RES_LAUNCH
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : int(e);
}

}  // namespace bnmi

#ifdef BN_TILE_CLOCK
extern "C" int bn_debug_tile_clock_resident(unsigned long long* out, int n_tiles) {
    if (n_tiles > bnmi::kTileClockTiles) n_tiles = bnmi::kTileClockTiles;
    return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(bnmi::g_tile_clock), sizeof(unsigned long long) * bnmi::kTileClockStamps * n_tiles));
}
#endif

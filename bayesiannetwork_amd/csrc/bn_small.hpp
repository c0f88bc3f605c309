// bn_small.hpp -- SMALL networks (ALARM-sized: what the reference's users actually load): the whole run in ONE
// workgroup with the complete state -- messages, node vectors, CPT, staged terms -- in the CU's LDS (bn_small.hip).
//
// The tile layout (bn_plan.hpp) gives a wavefront to a handful of nodes; on a network of 37 nodes that is 21
// wavefronts of ~2 000 instructions each, one launch per iteration: ~10 us per sweep whatever the size.  Here the
// work items are the things the reference's loops enumerate (belief_propagation.hpp):
//   entry item        one CPT entry (node v, parent assignment a, own state i): its term of pi(v)[i] (:174-200) and
//                     of the lambda-message to every parent (:240-266), written to the PLACE the term has in its
//                     accumulator's summation order;
//   accumulator item  one element of pi(v) or of a lambda-message: adds its run of staged terms front to back (the
//                     reference's order: own state outer, assignment inner), normalises (:298-311), residual;
//   product item      one element of lambda(v) (:220-238) or of a pi-message (:202-218): product over the children
//                     in ascending order, normalise, residual.
// The elements of one vector sit in adjacent lanes of one wavefront (normalisation never crosses a wave); two
// workgroup barriers per iteration; the stop decision (:147) is taken by every wave from the same LDS words.
// Sums and products keep the reference's order for ANY table size: results equal the CPU restatement bit for bit
// (the reference itself is order-nondeterministic only in the product over >= 3 parents, :253 -- ascending here).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "bn_device.hpp"

namespace bnmi {

constexpr int kSmallMaxWaves = 16;       // one workgroup of up to 1024 threads
constexpr int kSmallPreferredWaves = 12; // ... of which three per SIMD run faster than four (bn_small_plan.cpp)
constexpr int kSmallMaxRounds = 4;       // items of one kind a thread handles at most
constexpr int kSmallMaxParents = 8;
constexpr int kSmallLdsBytes = 150 * 1024;
constexpr int kSmallBudget = 1 << 16;    // iterations per launch

struct SmallSlot { uint32_t x, y, z, w; };  // one accumulator / product item (bn_small_plan.cpp: encodings)
struct SmallEntry { uint32_t x, y; };       // one entry item

// Host plan + its device image.
struct SmallPlan {
    bool ok = false;
    std::string why;                 // not eligible because ...
    int32_t n = 0, N = 0, M = 0;     // nodes, sum of arities, sum over edges of the parent's arity
    int32_t S = 0, T = 0, TT = 0;    // CPT entries, staged terms (doubles), per-entry parent terms (u32)
    int32_t CL = 0;                  // child-list entries
    int32_t waves = 0, re = 0, rb = 0, rc = 0;  // workgroup size, rounds per item kind
    int32_t mmax = 0;
    size_t lds_bytes = 0;
    std::vector<SmallEntry> ent;     // [re][threads]
    std::vector<double> ent_cpt;     // [re][threads]
    std::vector<uint32_t> term;      // [TT]   gather index (pi-message element) | staging place of the lambda term << 16
    std::vector<uint16_t> clist;     // [CL]   first element of the child's lambda-message
    std::vector<SmallSlot> bslot;    // [rb][threads] accumulator items
    std::vector<SmallSlot> cslot;    // [rc][threads] product items
    std::vector<int32_t> nv_idx;     // [N]    where the evidence kernel puts element x of a node vector (doubles into node0)
    std::vector<int32_t> nv_slot;    // [N]    the node's evidence-mark slot
    std::vector<double> npi_init;    // [N]    initial pi(v): the CPT row of a root, else 1.0 (:38-64)
    std::vector<int32_t> node_off;   // [n + 1] first element of node v's vectors
    int32_t v0 = 0, v1 = 0;          // the nodes whose items these are (the whole network unless a part of a MidPlan)
};

// A network spread over several workgroups (bn_mid.hip): one SmallPlan per contiguous node range, message / node-vector indices
// global (the state lives in memory, exchanged through L2 with agent-scope accesses and a grid barrier per iteration).
constexpr int kMidMaxParts = 224;   // workgroups of one run: all co-resident, one per CU (the engine admits a plan only below 0.9 x CUs); <= 4 x 64 flags per barrier poll
constexpr int kMidSyncBytes = 128 + 128 * kMidMaxParts;   // per state slot: 128 bytes (unused), then two granule tables of 16 bytes per workgroup (bn_mid.hip mid_grid_barrier); sized for the
                                                          // flag-per-line form of rounds 3-4
constexpr int kMidPreferredParts = 96;   // parts of at least 2 000 staged terms, about this many where the network is large enough (measured, bn_small_plan.cpp)
struct MidPlan {
    bool ok = false;
    std::string why;
    std::vector<SmallPlan> parts;
    int32_t waves = 0, rounds = 0;   // the largest of the parts': one launch configuration for all workgroups
    size_t lds_bytes = 0;
    int64_t est_total = 0;           // estimated staged terms of the whole network (the planner's measure of size)
};
void build_mid_plan(const Plan& p, MidPlan& mp);

// Builds the plan from the model held in `p` (single rank).  sp.ok false + sp.why when the network does not fit.
void build_small_plan(const Plan& p, SmallPlan& sp);

struct SmallArgs {
    BpBuffers b;              // evidence marks, node0 (evidence vectors), beliefs, res_hist
    double eps;
    int32_t max_sweeps, sweep_begin, budget;
    uint32_t run_id;
    Ctl* host_ctl;
    int32_t n, N, M, S, T, TT, CL, re, rb, rc, mmax;
    const SmallEntry* ent;
    const double* ent_cpt;
    const uint32_t* term;
    const uint16_t* clist;
    const SmallSlot* bslot;
    const SmallSlot* cslot;
    const int32_t* nv_idx;
    const int32_t* nv_slot;
    const double* npi_init;
    // Evidence (:68-73).  ev_mode 0: the marks and vectors bp_evidence_kernel left in the tile buffers (b.frozen, b.node0);
    // 1: the caller's arrays, read in place -- no evidence launch in front of the run: set q has ev_meta[8 q .. 8 q + 4] =
    // {count, first node entry, first offset entry, first value, number of values} (ev_meta null: one set, ev_ne entries and
    // ev_nval values from the start)
    int32_t ev_mode, ev_ne, ev_nval;
    const int32_t* ev_node;
    const int32_t* ev_off;
    const double* ev_val;
    const int32_t* ev_meta;
    const int32_t* node_off;
    double* state;            // [2M + 2N] pi-messages, lambda-messages (CSR edge order), pi(v), lambda(v): the state the
                              // launch stopped in (bn_bp_messages; a run longer than one launch's budget continues from it)
    // several evidence sets, one workgroup each (blockIdx.x): strides of the per-set arrays
    SetStrides sets;
    int64_t state_stride;
};
// ---- bn_mid.hip: several workgroups, state in memory
struct MidPart {          // one workgroup's share (device array [nparts])
    int32_t v0, v1;       // its nodes
    int32_t ent_off, term_off, clist_off, bslot_off, cslot_off;  // where its tables start in the concatenated arrays
    int32_t re, rb, rc, T, TT, CL, nt;                           // rounds per item kind, staged terms, parent terms, child-list entries, threads
};
struct MidArgs {
    BpBuffers b;              // evidence marks, node0 (evidence vectors), beliefs, res_hist
    double eps;
    int32_t max_sweeps, sweep_begin, budget;
    uint32_t run_id;
    Ctl* host_ctl;
    int32_t n, N, M, nparts;
    const MidPart* parts;
    const SmallEntry* ent;
    const double* ent_cpt;
    const uint32_t* term;
    const uint16_t* clist;
    const SmallSlot* bslot;
    const SmallSlot* cslot;
    const int32_t* nv_idx;
    const int32_t* nv_slot;
    const double* npi_init;
    const int32_t* node_off;  // [n + 1]
    const int32_t* msg_first; // [n + 1] first message element of node v's in-edges
    int32_t ev_mode, ev_ne;   // as SmallArgs (ev_meta: this set's header or null)
    const int32_t* ev_node;
    const int32_t* ev_off;
    const double* ev_val;
    const int32_t* ev_meta;
    // the state, in device memory: [2][M] pi-messages, [2][M] lambda-messages (CSR edge order), [2][N] pi(v), [2][N] lambda(v), marks
    double* pi;
    double* lam;
    double* npi;
    double* nlam;
    uint8_t* frz;
    unsigned* bar;               // the slot's barrier words, zeroed by the host before a launch: from byte 128 on two tables (by the parity of the
                                 // barrier's number) of one 16-byte granule per workgroup, {number | residual high half}, {number | residual low half}
    unsigned long long* res;     // (unused since round 5: maximum_difference travels in the granules)
    unsigned* abort;             // page-locked host word
    unsigned long long timeout_ticks;
    int32_t first_poll_delay;    // 10 ns ticks between the predicted arrival of the last workgroup and a workgroup's first poll (-1: poll from the own arrival on)
    // several evidence sets in one launch (gridDim.y): set blockIdx.y of the launch is set `set_base + blockIdx.y` of the batch
    // (its beliefs, residual history, control block, evidence header) and works in state slot `slot_base + blockIdx.y`
    // (state, marks and barrier words, `state_stride` doubles / N bytes / 64 bytes apart)
    SetStrides sets;
    int32_t set_base, slot_base;
    int64_t state_stride;
};
int prepare_bp_mid();
int launch_bp_mid(const MidArgs& a, int waves, int rounds, size_t lds_bytes, int n_sets, void* stream);

int prepare_bp_small();  // once per device, before the first launch
int launch_bp_small(const SmallArgs& a, int waves, size_t lds_bytes, int n_sets, void* stream);

}  // namespace bnmi

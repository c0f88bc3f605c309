// bn_lw.hpp -- likelihood weighting on the GPU (reference: bayesian/inference/likelihood_weighting.hpp).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "bn_plan.hpp"

namespace bnmi {

constexpr int kLwThreads = 256;      // threads per block
#ifndef BN_LW_PER_THREAD
#define BN_LW_PER_THREAD 4
#endif
constexpr int kLwPerThread = BN_LW_PER_THREAD;      // samples per thread
constexpr int kLwBlockSamples = kLwThreads * kLwPerThread;

// One topological position: everything a wave needs about the node, 32 bytes, wave-uniform, read
// with one scalar load a position ahead.  The first four parents are inline (node | arity << 24,
// 0 = none) when every node id fits 24 bits (LwState::inline_parents); longer lists and larger
// networks use the LwParent list.
struct LwStep {
    uint32_t coff_lo;  // offset of the node's CPT in the flat array (doubles), 48 bits
    int32_t v;         // node id (row of the state matrix)
    uint32_t par_off;  // first entry of the node's parents in LwParent[] (even); kLwStepPacked steps (<= 4 parents, all inline): first row of the node's table in LwState::d_thr16 (even)
    uint16_t coff_hi;
    uint8_t kv, m;     // arity; number of parents (low five bits) | kLwStepPacked | kLwStepPow2
    uint32_t par[4];   // parents 0..3: node | arity << 24
};
constexpr int kLwStepPacked = 0x80;  // <= 4 parents and <= 256 CPT rows: the kernel keeps four samples' row numbers in the bytes of one register
constexpr int kLwStepPow2 = 0x40;    // ... and every parent's arity is a power of two: shifts instead of multiplies
// lw_sample_small_kernel's descriptor of one topological position: 64 bytes, wave-uniform, one scalar load a position ahead.  Everything a
// position addresses is in it READY TO USE -- row addresses, the table's buffer descriptor: the kernel is bound by scalar issue (a SIMD
// issues one scalar instruction per four cycles, like one vector instruction; round 5 counted ~108 scalar against ~80 vector
// instructions per position, 25 of the former 64-bit address arithmetic on values that never change between launches).
struct LwSmallStep {
    uint64_t par[4];   // ADDRESS of each parent's row in the state matrix (states + node x row stride); a missing parent: row n, all zero
    uint64_t own;      // ... of the node's own row
    uint32_t tab[4];   // buffer descriptor of the node's table in LwState::d_thr16: base, num_records = its bytes (8 per row, rounded
                       // up to 16): what the wave copies to LDS -- lanes beyond the table read zeros and request nothing
    uint32_t coff;     // offset of the node's CPT / thresholds in the flat arrays (entries; below 2^32 on this path)
    uint32_t shape;    // arity | parents << 4 | a1 << 8 | a2 << 16 | a3 << 24: parents 1..3's arities (small_pow2: log2 of them); a missing parent: arity 1;
                       // bit 7: parent j (bits 12-13) is the node of the position before -- its byte is taken from that position's store, not from memory
};
struct LwParent {
    uint32_t node, k;  // parent node id and its arity (mixed-radix digit base)
};

struct LwState {
    bool ready = false;
    int32_t* d_k = nullptr;
    int64_t* d_node_off = nullptr;
    double* d_cpt = nullptr;       // flat, reference row order (row lookup = k contiguous doubles)
    unsigned long long* d_thr = nullptr;  // same layout: entry i of a row = ceil(running total up to state i x 2^53), the selection thresholds (bn_lw_kernels.hip pick_states)
    uint32_t* d_thr32 = nullptr;   // ... and their top halves (threshold >> 21): what a draw is compared with first, 4 bytes per entry
    uint32_t* d_thr16 = nullptr;   // nodes with <= 256 rows (kLwStepPacked): per row 8 bytes {t0 | t1 << 16, t2 | 0xffff << 16}, t = threshold >> 37 -- the copy a wave stages in LDS
    LwStep* d_steps = nullptr;
    LwSmallStep* d_small_steps = nullptr;   // [n + 3] in topological order (LwState::small), written for the state matrix in use (d_states, batch)
    std::vector<LwSmallStep> h_small;       // the same with NODE numbers in par / own and {first row in d_thr16, bytes} in tab[0..1]: what the device copy is made from
    LwParent* d_parents = nullptr; // [E] grouped by position, first parent first
    int32_t* d_ev_topo = nullptr;  // [n] clamped state or -1 of the node at each position
    int32_t kmax = 0;              // largest arity
    bool rows24 = false;           // every CPT has < 2^24 rows: 24-bit row arithmetic
    bool inline_parents = false;   // n <= 2^24: LwStep::par is filled
    bool small = false;            // every node has <= 4 parents, <= 256 CPT rows and <= 4 states (and n < 2^24 - 1): lw_sample_small_kernel; LwStep::par of a
                                   // missing parent then names the all-zero row n of the state matrix with arity 1
    bool small_pow2 = false;       // ... and every arity is a power of two: LwStep::par holds log2 of the arity in its top byte
    uint8_t* d_states = nullptr;   // sampled states of the current batch, [n + 1] rows of `stride` bytes: one byte per sample, or -- LwState::small,
                                   // every arity <= 4 -- FOUR samples per byte, two bits each (sample s: bits 2 (s & 3) of byte s >> 2)
    uint64_t stride = 0;           // bytes per row of d_states: batch (or batch / 4) + 33 x 128
    double* d_weights = nullptr;   // [batch]
    double* d_hist = nullptr;      // [sum k]
    int32_t* h_ev = nullptr;       // page-locked staging of d_ev_topo: the upload needs no synchronisation of its own
    double* h_hist = nullptr;      // page-locked landing place of the histogram
    uint64_t batch = 0;            // samples a row of d_states holds (multiple of kLwBlockSamples)
    uint64_t launch_samples = 0;   // samples per launch of the current call (<= batch, the row stride)
    uint64_t last_batch_samples = 0;
    std::vector<int32_t> topo;
};

struct LwArgs {
    int32_t n;
    int32_t kmax;
    bool rows24;
    bool inline_parents;
    bool small, small_pow2;
    const LwStep* steps;
    const LwSmallStep* small_steps;
    const LwParent* parents;
    const int32_t* ev_topo;
    const int32_t* k;
    const int64_t* node_off;
    const double* cpt;
    const unsigned long long* thr;
    const uint32_t* thr32;
    const uint32_t* thr16;
    uint8_t* states;
    double* weights;
    double* hist;
    uint64_t batch;        // row stride of `states` in bytes
    bool packed2;          // `states` holds four samples per byte, two bits each (the straight-line kernel's networks: every arity <= 4)
    uint64_t sample_base;  // global index of the batch's sample 0
    uint64_t n_valid;      // samples of this batch that count
    uint64_t seed;
    int32_t mode;          // 0 likelihood weighting, 1 rejection (logic) sampling
};

int launch_lw_sample(const LwArgs& a, int blocks, void* stream);
int launch_lw_hist(const LwArgs& a, int blocks, void* stream);
int launch_lw_transpose(const uint8_t* states, uint8_t* out, int32_t n, uint64_t stride, bool packed2, uint64_t n_samples, void* stream);

void lw_free(LwState& s);
// hist_out == nullptr: leave the histogram in s.d_hist (the caller reduces it across ranks first)
int lw_run(LwState& s, const Plan& p, void* stream, int32_t ne, const int32_t* ev_node, const int32_t* ev_state,
           uint64_t sample_begin, uint64_t n_samples, uint64_t seed, double* hist_out, std::string& err);
// Rejection sampling: draw until n_accept samples agree with the evidence (at most max_draw draws);
// counts_out = un-normalised state counts of the first n_accept accepted samples.
int rs_run(LwState& s, const Plan& p, void* stream, int32_t ne, const int32_t* ev_node, const int32_t* ev_state,
           uint64_t sample_begin, uint64_t n_accept, uint64_t max_draw, uint64_t seed, double* counts_out,
           uint64_t* drawn_out, uint64_t* accepted_out, std::string& err);
int lw_states(LwState& s, const Plan& p, void* stream, uint64_t n, uint8_t* states_out, double* weights_out,
              std::string& err);

}  // namespace bnmi

/*
 * oracle/lw_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's Likelihood Weighting sampler
 * (bayesian/inference/likelihood_weighting.hpp) over the flat model of bp_oracle.c.
 *
 * What is restated exactly:
 *   weighted_sample   :122-173  ancestral sampling, evidence nodes clamp and multiply
 *                               the weight by their CPT entry
 *   make_random_by_weight :177-193  first i with cum_{i-1} <= u < cum_i, else last state
 *   accumulate        :33-50    hist[v][state_v] += w for every node of every sample
 *   normalize         :197-221  divide by the sum; uniform when the sum < 1e-20
 * What is NOT the reference's: the random stream.  The reference draws from an mt19937
 * seeded by std::random_device (:224-244) -- non-deterministic by design, so no bitwise
 * parity exists even reference-vs-reference.  This repository's sampler (oracle and HIP
 * kernel alike) gives every sample its own xoshiro128++ stream (Blackman & Vigna 2019),
 * seeded by one block of the counter-based Philox4x32-10 generator (Salmon et al., SC'11):
 *   state(sample s) = philox4x32_10(counter = {s_lo, s_hi, 0, 0}, key = {seed_lo, seed_hi})
 *                     (an all-zero state, which xoshiro cannot leave, becomes {1,0,0,0})
 *   the stream takes ONE step per TWO topological positions (evidence nodes or not): at every EVEN
 *   position t = 0, 2, ...  out = next()  (the ++ output, 32 bits), and
 *       h(t) = out >> 16,   h(t + 1) = out & 0xffff            the top 16 bits of each position's uniform
 *       low(t)     = ss(x[1]) << 5 | ss(x[2]) >> 27             the 37 bits below them: the ** scrambler
 *       low(t + 1) = ss(x[3]) << 5 | ss(x[0]) >> 27             ss(w) = rotl(w * 5, 7) * 9 of words of the
 *                                                              state the step left behind
 *       U = h << 37 | low  (53 bits);   u(s, t) = U * 2^-53
 *   A draw is decided by its top 16 bits unless they equal the top 16 bits of a running total (3 x 2^-16
 *   per draw); only then are the low 37 looked at, and they come from state words the output function
 *   did not use.  (Round 4: one step per position, top 32 bits from the output.  The ten operations of a
 *   step were a fifth of the sampling kernel's vector instructions, and the kernel is bound by exactly
 *   those; 16 bits decide a draw as well as 32 do, so one output serves two positions.)
 * so u(s, t) depends on (seed, s, t) only -- not on the evidence set, the batch or the GPU --
 * which makes sampled STATES bit-reproducible between this file and the HIP kernel.
 * Parity with the reference itself is statistical (tests/golden holds the reference's
 * 1e5-sample marginals under a reseeded mt19937 plus exact BP marginals on polytrees).
 * Nodes are visited in ascending topological position `topo` (any topological order
 * yields the reference's distribution; the reference's own order is a DFS from the
 * last vertex, :162-170).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

/* Philox4x32-10, Random123 definition. */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0, p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* xoshiro128++ 1.0 (Blackman & Vigna), 32-bit output, state x[4]. */
static uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
uint32_t oracle_xoshiro128pp_next(uint32_t x[4]) {
    uint32_t result = rotl32(x[0] + x[3], 7) + x[0];
    uint32_t t = x[1] << 9;
    x[2] ^= x[0]; x[3] ^= x[1]; x[1] ^= x[2]; x[0] ^= x[3];
    x[2] ^= t;
    x[3] = rotl32(x[3], 11);
    return result;
}

static void stream_seed(uint64_t seed, uint64_t s, uint32_t x[4]) {
    uint32_t ctr[4] = {(uint32_t)s, (uint32_t)(s >> 32), 0u, 0u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    oracle_philox4x32_10(ctr, key, x);
    if ((x[0] | x[1] | x[2] | x[3]) == 0) x[0] = 1;
}

/* the ** scrambler of xoshiro128 on one word of the state as it stands (no step) */
static uint32_t ss32(uint32_t w) { return rotl32(w * 5u, 7) * 9u; }
uint32_t oracle_xoshiro128ss_peek(const uint32_t x[4]) { return ss32(x[1]); }

/* u(s, t) for t = 0, 1, 2, ... in order: *cur carries the output of the even position's step to the odd one */
double oracle_lw_stream_uniform(uint32_t x[4], uint32_t *cur, uint32_t t) {
    uint32_t h;
    uint64_t low;
    if ((t & 1u) == 0) {
        *cur = oracle_xoshiro128pp_next(x);
        h = *cur >> 16;
        low = ((uint64_t)ss32(x[1]) << 5) | (ss32(x[2]) >> 27);
    } else {
        h = *cur & 0xffffu;
        low = ((uint64_t)ss32(x[3]) << 5) | (ss32(x[0]) >> 27);
    }
    uint64_t v = ((uint64_t)h << 37) | low;
    return (double)v * (1.0 / 9007199254740992.0);
}
void oracle_lw_stream_seed(uint64_t seed, uint64_t s, uint32_t x[4]) { stream_seed(seed, s, x); }

/* the uniform of (sample s, position t): test hook */
double oracle_lw_uniform(uint64_t seed, uint64_t s, uint32_t t) {
    uint32_t x[4], cur = 0;
    stream_seed(seed, s, x);
    double u = 0;
    for (uint32_t i = 0; i <= t; ++i) u = oracle_lw_stream_uniform(x, &cur, i);
    return u;
}

/* likelihood_weighting.hpp:177-193 */
static int pick_state(double u, const double *w, int k) {
    double total = 0.0;
    for (int i = 0; i < k; ++i) {
        double old_total = total;
        total += w[i];
        if (old_total <= u && u < total) return i;
    }
    return k - 1;
}

/*
 * Draw samples [s_begin, s_begin + n_samples) and accumulate the UN-normalised weighted
 * histogram into hist (sum_v k[v] doubles, node-major; must be zeroed by the caller).
 * ev_state[v] = clamped state or -1.  topo[t] = node visited at position t.
 * states_out (optional) receives the sampled states of the first states_cap samples,
 * sample-major [s][v] as uint8.  weights_out (optional) the first states_cap weights.
 */
int oracle_lw_run(int n, const int32_t *k, const int32_t *in_ptr, const int32_t *in_idx,
                  const int64_t *cpt_off, const double *cpt, const int32_t *topo,
                  const int32_t *ev_state, uint64_t s_begin, uint64_t n_samples, uint64_t seed,
                  double *hist, uint8_t *states_out, double *weights_out, uint64_t states_cap) {
    int64_t *node_off = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
    int32_t *state = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    if (!node_off || !state) { free(node_off); free(state); return -1; }
    node_off[0] = 0;
    for (int v = 0; v < n; ++v) node_off[v + 1] = node_off[v] + k[v];
    for (uint64_t si = 0; si < n_samples; ++si) {
        uint64_t s = s_begin + si;
        double w = 1.0; /* :124 */
        uint32_t rng[4], cur = 0;
        stream_seed(seed, s, rng);
        for (int t = 0; t < n; ++t) {
            int v = topo[t];
            double u = oracle_lw_stream_uniform(rng, &cur, (uint32_t)t); /* every position has its uniform, evidence node or not */
            int64_t row = 0; /* parent assignment -> CPT row, first parent most significant */
            for (int32_t e = in_ptr[v]; e < in_ptr[v + 1]; ++e) row = row * k[in_idx[e]] + state[in_idx[e]];
            const double *r = cpt + cpt_off[v] + row * k[v];
            if (ev_state[v] >= 0) { /* :148-153 */
                w *= r[ev_state[v]];
                state[v] = ev_state[v];
            } else { /* :154-158 */
                state[v] = pick_state(u, r, k[v]);
            }
        }
        for (int v = 0; v < n; ++v) hist[node_off[v] + state[v]] += w; /* :45-49 */
        if (si < states_cap) {
            if (states_out) for (int v = 0; v < n; ++v) states_out[si * (uint64_t)n + v] = (uint8_t)state[v];
            if (weights_out) weights_out[si] = w;
        }
    }
    free(node_off); free(state);
    return 0;
}

/* likelihood_weighting.hpp:197-221: normalise one node's histogram in place. */
void oracle_lw_normalize(double *h, int k) {
    double sum = 0;
    for (int i = 0; i < k; ++i) sum += h[i];
    if (sum < 1.0e-20) for (int i = 0; i < k; ++i) h[i] = 1.00 / k;
    else for (int i = 0; i < k; ++i) h[i] /= sum;
}

/*
 * Rejection (logic) sampling, reference rejection_sampling.hpp:33-167: every node -- evidence
 * nodes too -- is sampled (choice_pattern, :115-167), a sample counts only when it agrees with
 * every condition (:70-84), sampling goes on until n_accept samples were accepted (:93-111;
 * bounded here by max_draw), and the marginals are plain counts over the accepted samples
 * (:40-58).  Same random stream and topological walk as oracle_lw_run, so the accepted set is
 * bit-identical to the HIP path's.  counts must be zeroed by the caller.
 */
int oracle_rs_run(int n, const int32_t *k, const int32_t *in_ptr, const int32_t *in_idx,
                  const int64_t *cpt_off, const double *cpt, const int32_t *topo, const int32_t *ev_state,
                  uint64_t s_begin, uint64_t n_accept, uint64_t max_draw, uint64_t seed, double *counts,
                  uint64_t *drawn_out, uint64_t *accepted_out) {
    int64_t *node_off = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
    int32_t *state = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    if (!node_off || !state) { free(node_off); free(state); return -1; }
    node_off[0] = 0;
    for (int v = 0; v < n; ++v) node_off[v + 1] = node_off[v] + k[v];
    uint64_t drawn = 0, accepted = 0;
    while (accepted < n_accept && drawn < max_draw) {
        uint64_t s = s_begin + drawn;
        int ok = 1;
        uint32_t rng[4], cur = 0;
        stream_seed(seed, s, rng);
        for (int t = 0; t < n; ++t) {
            int v = topo[t];
            double u = oracle_lw_stream_uniform(rng, &cur, (uint32_t)t);
            int64_t row = 0;
            for (int32_t e = in_ptr[v]; e < in_ptr[v + 1]; ++e) row = row * k[in_idx[e]] + state[in_idx[e]];
            state[v] = pick_state(u, cpt + cpt_off[v] + row * k[v], k[v]);
            if (ev_state[v] >= 0 && state[v] != ev_state[v]) ok = 0;
        }
        ++drawn;
        if (ok) {
            ++accepted;
            for (int v = 0; v < n; ++v) counts[node_off[v] + state[v]] += 1.0;
        }
    }
    if (drawn_out) *drawn_out = drawn;
    if (accepted_out) *accepted_out = accepted;
    free(node_off); free(state);
    return 0;
}

/*
 * CPT fitting, restated from bayesian/sampler.hpp:81-163 (sampler::make_cpt).  PARITY UNPINNED for
 * this function: sampler.hpp needs Boost, which this image lacks, so the reference side cannot be
 * compiled here; the restatement follows the source text: per node and parent assignment the
 * occurrence counts of the patterns are summed by own state (:96-122); a row is count / row total
 * (:148-151), or uniform when the total is zero (:140-146).
 */
int oracle_make_cpt(int n, const int32_t *k, const int32_t *in_ptr, const int32_t *in_idx,
                    const int64_t *cpt_off, int64_t n_patterns, const uint8_t *patterns /* [P][n] */,
                    const uint64_t *counts, double *cpt_out) {
    int64_t total = cpt_off[n];
    uint64_t *cnt = (uint64_t *)calloc((size_t)(total ? total : 1), sizeof(uint64_t));
    if (!cnt) return -1;
    for (int64_t p = 0; p < n_patterns; ++p)
        for (int v = 0; v < n; ++v) {
            int64_t row = 0;
            for (int32_t e = in_ptr[v]; e < in_ptr[v + 1]; ++e) row = row * k[in_idx[e]] + patterns[p * n + in_idx[e]];
            cnt[cpt_off[v] + row * k[v] + patterns[p * n + v]] += counts[p];
        }
    for (int v = 0; v < n; ++v)
        for (int64_t o = cpt_off[v]; o < cpt_off[v + 1]; o += k[v]) {
            uint64_t s = 0;
            for (int i = 0; i < k[v]; ++i) s += cnt[o + i];
            double parameter = (double)s;
            for (int i = 0; i < k[v]; ++i) cpt_out[o + i] = (s == 0) ? 1.0 / k[v] : (double)cnt[o + i] / parameter;
        }
    free(cnt);
    return 0;
}

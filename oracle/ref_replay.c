/*
 * oracle/ref_replay.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Replay oracle for the reference's sampler family: restates, draw for draw, what
 *   likelihood_weighting::operator()      bayesian/inference/likelihood_weighting.hpp:28-59
 *   likelihood_weighting::make_samples    :62-117
 *   likelihood_weighting::weighted_sample :122-173   (visiting order :162-170, recursion :133-145)
 *   make_random_by_weight                 :177-193
 *   normalize                             :197-221
 *   rejection_sampling::operator()        bayesian/inference/rejection_sampling.hpp:33-62
 *   generate_pattern / choice_pattern     :65-167
 * do once their probability_generator (:224-244 / :170-190) holds a std::mt19937 with a KNOWN seed.
 * The reference seeds that engine from std::random_device, so a reference run is reproducible only
 * after reseeding it (oracle/ref_driver.cpp does that through -Dprivate=public); from then on the
 * run is fully deterministic and this file reproduces its outputs BIT FOR BIT -- that is what pins
 * SURVEY rows A13-A17 / f-2 / f-4 to the reference (tests/test_oracle_golden.py compares against
 * tests/golden/lw_*.npz, ms_*.npz, rs_*.npz, all produced by the reference itself).
 *
 * Pieces that are not in /root/reference but in its toolchain (libstdc++ 11, the reference's only
 * external dependency on this path), restated from their published definitions:
 *   std::mt19937                       Matsumoto & Nishimura 1998 (MT19937, 32-bit), seeding
 *                                      x[i] = 1812433253 * (x[i-1] ^ (x[i-1] >> 30)) + i
 *   std::uniform_real_distribution<double>(0,1)(engine)
 *                                      = generate_canonical<double,53>: two engine words,
 *                                        (lo + hi * 2^32) / 2^64 evaluated in double, clamped below 1
 *
 * The same walk can also be driven by this repository's own random stream (Philox-seeded
 * xoshiro128++ per sample id, one uniform per topological position; see lw_oracle.c) so that the
 * GPU path's adaptive-stop loop (units executed, pattern table) has an oracle fed with the GPU's
 * stream: stream_kind 1.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint32_t oracle_xoshiro128pp_next(uint32_t x[4]);
uint32_t oracle_xoshiro128ss_peek(const uint32_t x[4]);

/* ---- std::mt19937 ------------------------------------------------------------------------ */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;

static void mt_seed(mt19937_t *g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static uint32_t mt_next(mt19937_t *g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* libstdc++ generate_canonical<double, 53, mt19937>, then (b - a) * x + a with a = 0, b = 1 */
static double mt_uniform01(mt19937_t *g) {
    double sum = 0.0, tmp = 1.0;
    for (int kk = 2; kk != 0; --kk) {
        sum += (double)mt_next(g) * tmp;
        tmp *= 4294967296.0;
    }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret * (1.0 - 0.0) + 0.0;
}

/* test hooks */
void oracle_mt19937_words(uint32_t seed, int n, uint32_t *out) {
    mt19937_t g;
    mt_seed(&g, seed);
    for (int i = 0; i < n; ++i) out[i] = mt_next(&g);
}
void oracle_mt19937_uniforms(uint32_t seed, int n, double *out) {
    mt19937_t g;
    mt_seed(&g, seed);
    for (int i = 0; i < n; ++i) out[i] = mt_uniform01(&g);
}

/* ---- visiting order ---------------------------------------------------------------------- */
/* weighted_sample :162-170 / generate_pattern :93-103: take the LAST remaining vertex; before a
 * vertex is sampled every parent that is still remaining is taken out and sampled first, parents
 * in in_vertexes order = ascending vertex_list() position (graph.hpp:405-433).  The order does not
 * depend on the sample, so it is computed once. */
static void visit(int v, const int32_t *in_ptr, const int32_t *in_idx, uint8_t *remaining, int32_t *order, int *cnt) {
    for (int32_t e = in_ptr[v]; e < in_ptr[v + 1]; ++e) {
        int p = in_idx[e];
        if (remaining[p]) {
            remaining[p] = 0;
            visit(p, in_ptr, in_idx, remaining, order, cnt);
        }
    }
    order[(*cnt)++] = v;
}

int oracle_ref_visit_order(int n, const int32_t *in_ptr, const int32_t *in_idx, int32_t *order_out) {
    uint8_t *remaining = (uint8_t *)malloc((size_t)n + 1);
    if (!remaining) return -1;
    memset(remaining, 1, (size_t)n + 1);
    int cnt = 0;
    for (int v = n - 1; v >= 0; --v)
        if (remaining[v]) {
            remaining[v] = 0;
            visit(v, in_ptr, in_idx, remaining, order_out, &cnt);
        }
    free(remaining);
    return cnt == n ? 0 : -2;
}

/* ---- one sample -------------------------------------------------------------------------- */
typedef struct {
    int n;
    const int32_t *k, *in_ptr, *in_idx;
    const int64_t *cpt_off;
    const double *cpt;
    const int32_t *order;     /* visiting order */
    int stream_kind;          /* 0: mt19937, evidence nodes draw nothing; 1: repo stream, one draw per position */
    mt19937_t mt;
    uint64_t seed;
} walker_t;

static const double *row_of(const walker_t *w, int v, const int32_t *state) {
    int64_t row = 0; /* parent assignment -> CPT row, first parent most significant (A0 / A12) */
    for (int32_t e = w->in_ptr[v]; e < w->in_ptr[v + 1]; ++e) row = row * w->k[w->in_idx[e]] + state[w->in_idx[e]];
    return w->cpt + w->cpt_off[v] + row * w->k[v];
}

/* the repository's stream: defined in lw_oracle.c (one xoshiro128++ step per two positions) */
void oracle_lw_stream_seed(uint64_t seed, uint64_t s, uint32_t x[4]);
double oracle_lw_stream_uniform(uint32_t x[4], uint32_t *cur, uint32_t t);

/* likelihood_weighting.hpp:177-193 */
static int make_random_by_weight(double value, const double *weight, int k) {
    double total = 0.0;
    for (int i = 0; i < k; ++i) {
        double old_total = total;
        total += weight[i];
        if (old_total <= value && value < total) return i;
    }
    return k - 1;
}

/* weighted_sample :122-173; sample_id is used by the repo stream only */
static double weighted_sample(walker_t *w, const int32_t *ev_state, uint64_t sample_id, int32_t *state) {
    double weight = 1.0;
    uint32_t x[4], cur = 0;
    if (w->stream_kind == 1) oracle_lw_stream_seed(w->seed, sample_id, x);
    for (int t = 0; t < w->n; ++t) {
        int v = w->order[t];
        double u = 0.0;
        if (w->stream_kind == 1) u = oracle_lw_stream_uniform(x, &cur, (uint32_t)t); /* every position has its uniform */
        const double *r = row_of(w, v, state);
        if (ev_state[v] >= 0) { /* :148-153 */
            weight *= r[ev_state[v]];
            state[v] = ev_state[v];
        } else { /* :154-158 */
            if (w->stream_kind == 0) u = mt_uniform01(&w->mt);
            state[v] = make_random_by_weight(u, r, w->k[v]);
        }
    }
    return weight;
}

/* normalize :197-221 on one node's vector */
static void normalize_node(const double *src, double *dst, int k) {
    double sum = 0;
    for (int j = 0; j < k; ++j) sum += src[j];
    if (sum < 1.0e-20) for (int j = 0; j < k; ++j) dst[j] = 1.00 / k;
    else for (int j = 0; j < k; ++j) dst[j] = src[j] / sum;
}

static int walker_init(walker_t *w, int n, const int32_t *k, const int32_t *in_ptr, const int32_t *in_idx,
                       const int64_t *cpt_off, const double *cpt, const int32_t *order, int stream_kind, uint64_t seed,
                       int32_t **own_order) {
    w->n = n; w->k = k; w->in_ptr = in_ptr; w->in_idx = in_idx; w->cpt_off = cpt_off; w->cpt = cpt;
    w->stream_kind = stream_kind; w->seed = seed;
    *own_order = NULL;
    if (!order) {
        *own_order = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
        if (!*own_order || oracle_ref_visit_order(n, in_ptr, in_idx, *own_order)) return -1;
        order = *own_order;
    }
    w->order = order;
    if (stream_kind == 0) mt_seed(&w->mt, (uint32_t)seed);
    return 0;
}

/*
 * likelihood_weighting::operator()(evidence, sample_num) with the engine reseeded mt19937(mt_seed):
 * marg_out [sum k] = the reference's returned marginals.
 */
int oracle_ref_lw_run(int n, const int32_t *k, const int32_t *in_ptr, const int32_t *in_idx, const int64_t *cpt_off,
                      const double *cpt, const int32_t *ev_state, uint64_t sample_num, uint32_t mt_seed_value,
                      double *marg_out) {
    walker_t w;
    int32_t *own = NULL;
    int32_t *state = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    int64_t *node_off = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
    if (!state || !node_off || walker_init(&w, n, k, in_ptr, in_idx, cpt_off, cpt, NULL, 0, mt_seed_value, &own)) {
        free(state); free(node_off); free(own);
        return -1;
    }
    node_off[0] = 0;
    for (int v = 0; v < n; ++v) node_off[v + 1] = node_off[v] + k[v];
    double *ret = (double *)calloc((size_t)(node_off[n] ? node_off[n] : 1), sizeof(double)); /* :33-37 */
    if (!ret) { free(state); free(node_off); free(own); return -1; }
    for (uint64_t i = 0; i < sample_num; ++i) { /* :40-50 */
        double wt = weighted_sample(&w, ev_state, i, state);
        for (int v = 0; v < n; ++v) ret[node_off[v] + state[v]] += wt;
    }
    for (int v = 0; v < n; ++v) normalize_node(ret + node_off[v], marg_out + node_off[v], k[v]); /* :53-56 */
    free(ret); free(state); free(node_off); free(own);
    return 0;
}

/* ---- make_samples :62-117 ---------------------------------------------------------------- */
static int g_pat_n;
static int pat_cmp(const void *a, const void *b) { return memcmp(a, b, (size_t)g_pat_n); }

/*
 * order == NULL: the reference's visiting order.  stream_kind 0: mt19937(seed) (reference replay);
 * 1: the repository's stream, sample ids sample_begin, sample_begin + 1, ...
 * Outputs: units executed, the marginals of the last unit (`probabilities`), and the joint-pattern
 * table sorted lexicographically: patterns_out [<= pat_cap][n] states, counts_out [<= pat_cap];
 * n_patterns_out receives the number of distinct patterns (may exceed pat_cap: table truncated).
 * max_units bounds the loop (the reference has no bound); returns 1 if it was hit.
 */
int oracle_make_samples(int n, const int32_t *k, const int32_t *in_ptr, const int32_t *in_idx, const int64_t *cpt_off,
                        const double *cpt, const int32_t *order, const int32_t *ev_state, uint64_t unit_size,
                        double epsilon, uint64_t max_units, int stream_kind, uint64_t seed, uint64_t sample_begin,
                        uint64_t *units_out, double *marg_out, uint64_t pat_cap, uint8_t *patterns_out,
                        uint64_t *counts_out, uint64_t *n_patterns_out) {
    walker_t w;
    int32_t *own = NULL;
    int32_t *state = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    int64_t *node_off = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
    if (!state || !node_off || walker_init(&w, n, k, in_ptr, in_idx, cpt_off, cpt, order, stream_kind, seed, &own)) {
        free(state); free(node_off); free(own);
        return -1;
    }
    node_off[0] = 0;
    for (int v = 0; v < n; ++v) node_off[v + 1] = node_off[v] + k[v];
    const size_t hn = (size_t)(node_off[n] ? node_off[n] : 1);
    double *w_list = (double *)calloc(hn, sizeof(double));        /* :72-77 */
    double *prob = (double *)calloc(hn, sizeof(double));
    double *next = (double *)calloc(hn, sizeof(double));
    uint8_t *all = NULL;
    size_t all_cap = 0, all_cnt = 0;
    int rc = 0;
    uint64_t units = 0, drawn = 0;
    if (!w_list || !prob || !next) { rc = -1; goto done; }
    for (;;) {
        for (uint64_t i = 0; i < unit_size; ++i) { /* :82-99 */
            double wt = weighted_sample(&w, ev_state, sample_begin + drawn, state);
            ++drawn;
            for (int v = 0; v < n; ++v) w_list[node_off[v] + state[v]] += wt;
            if (all_cnt == all_cap) {
                all_cap = all_cap ? all_cap * 2 : 4096;
                uint8_t *na = (uint8_t *)realloc(all, all_cap * (size_t)(n ? n : 1));
                if (!na) { rc = -1; goto done; }
                all = na;
            }
            for (int v = 0; v < n; ++v) all[all_cnt * (size_t)n + v] = (uint8_t)state[v];
            ++all_cnt;
        }
        ++units;
        double max_difference = 2.2250738585072014e-308; /* numeric_limits<double>::min(), :101 */
        for (int v = 0; v < n; ++v) { /* :102-112 */
            normalize_node(w_list + node_off[v], next + node_off[v], k[v]);
            for (int j = 0; j < k[v]; ++j) {
                double d = fabs(prob[node_off[v] + j] - next[node_off[v] + j]);
                max_difference = (max_difference < d) ? d : max_difference; /* std::max */
            }
            for (int j = 0; j < k[v]; ++j) prob[node_off[v] + j] = next[node_off[v] + j];
        }
        if (max_difference < epsilon) break; /* :114 */
        if (max_units && units >= max_units) { rc = 1; break; }
    }
    if (units_out) *units_out = units;
    if (marg_out) memcpy(marg_out, prob, sizeof(double) * (size_t)node_off[n]);
    {   /* the pattern table (:93-98), as a sorted list */
        uint64_t distinct = 0;
        if (n > 0 && all_cnt > 0) {
            g_pat_n = n;
            qsort(all, all_cnt, (size_t)n, pat_cmp);
            size_t i = 0;
            while (i < all_cnt) {
                size_t j = i + 1;
                while (j < all_cnt && memcmp(all + i * (size_t)n, all + j * (size_t)n, (size_t)n) == 0) ++j;
                if (distinct < pat_cap && patterns_out && counts_out) {
                    memcpy(patterns_out + distinct * (size_t)n, all + i * (size_t)n, (size_t)n);
                    counts_out[distinct] = (uint64_t)(j - i);
                }
                ++distinct;
                i = j;
            }
        }
        if (n_patterns_out) *n_patterns_out = distinct;
    }
done:
    free(w_list); free(prob); free(next); free(all); free(state); free(node_off); free(own);
    return rc;
}

/* ---- rejection_sampling :33-167 ---------------------------------------------------------- */
/*
 * rejection_sampling::operator()(condition, generate_sample_num) with the engine reseeded
 * mt19937(mt_seed): every vertex is drawn (choice_pattern, :115-167, same visiting order), a pattern
 * is kept when it agrees with every condition pair (:70-84), until `num` were kept (:93-111);
 * marg_out = counts / num (:40-58).  cond_state[v] = required state or -1.
 * choice_pattern's selection loop has no upper bound (`for(int i = 0; current->selectable_num; ++i)`,
 * :153) and indexes the row with .at(i): when the uniform is not below the row total it throws
 * std::out_of_range -- reported here as return code 2.  max_draw bounds the loop (the reference has
 * no bound); return code 1 if it was hit.
 */
int oracle_ref_rs_run(int n, const int32_t *k, const int32_t *in_ptr, const int32_t *in_idx, const int64_t *cpt_off,
                      const double *cpt, const int32_t *cond_state, uint64_t num, uint32_t mt_seed_value,
                      uint64_t max_draw, double *marg_out, uint64_t *drawn_out) {
    walker_t w;
    int32_t *own = NULL;
    int32_t *state = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    int64_t *node_off = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
    if (!state || !node_off || walker_init(&w, n, k, in_ptr, in_idx, cpt_off, cpt, NULL, 0, mt_seed_value, &own)) {
        free(state); free(node_off); free(own);
        return -1;
    }
    node_off[0] = 0;
    for (int v = 0; v < n; ++v) node_off[v + 1] = node_off[v] + k[v];
    double *cnt = (double *)calloc((size_t)(node_off[n] ? node_off[n] : 1), sizeof(double));
    if (!cnt) { free(state); free(node_off); free(own); return -1; }
    uint64_t drawn = 0, kept = 0;
    int rc = 0;
    while (kept < num) {
        if (max_draw && drawn >= max_draw) { rc = 1; break; }
        for (int t = 0; t < n && rc == 0; ++t) {
            int v = w.order[t];
            const double *r = row_of(&w, v, state);
            double u = mt_uniform01(&w.mt); /* :150 */
            double total = 0.0;
            int pick = -1;
            for (int i = 0; i < k[v]; ++i) { /* :152-161 */
                double old_total = total;
                total += r[i];
                if (old_total <= u && u < total) { pick = i; break; }
            }
            if (pick < 0) rc = 2; /* .at(selectable_num) throws */
            state[v] = pick;
        }
        if (rc) break;
        ++drawn;
        int ok = 1; /* is_condition :70-84 */
        for (int v = 0; v < n; ++v)
            if (cond_state[v] >= 0 && state[v] != cond_state[v]) ok = 0;
        if (ok) {
            ++kept;
            for (int v = 0; v < n; ++v) cnt[node_off[v] + state[v]] += 1.0;
        }
    }
    if (rc == 0 || rc == 1)
        for (int64_t i = 0; i < node_off[n]; ++i) marg_out[i] = cnt[i] / (double)num; /* :52-55: divided by generated_patterns.size() */
    if (rc == 1) /* fewer than num kept: divide by what was kept, as .size() would be */
        for (int64_t i = 0; i < node_off[n]; ++i) marg_out[i] = kept ? cnt[i] / (double)kept : 0.0;
    if (drawn_out) *drawn_out = drawn;
    free(cnt); free(state); free(node_off); free(own);
    return rc;
}

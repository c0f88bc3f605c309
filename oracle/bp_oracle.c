/*
 * oracle/bp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's Loopy Belief
 * Propagation (bayesian/inference/belief_propagation.hpp) over the flat model
 * (CSR parents ascending + flat row-major CPT) used by this repository.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object; the product path (bayesiannetwork_amd/csrc)
 * never links or calls it.
 *
 * Parity status: PINNED.  oracle/ref_driver.cpp runs the unmodified reference
 * headers from /root/reference on the same flat models; tests/golden/ holds the
 * reference's outputs (marginals, sweep counts, per-sweep residuals) and
 * tests/test_oracle_golden.py checks this restatement against them, together
 * with the seven teacher vectors of libs/bayesian/test/belief_propagation.cpp.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference root).  Arithmetic is IEEE fp64 in the reference's operation order;
 * compile with -ffp-contract=off so no FMA is formed.
 *
 * Flat model:
 *   n            number of nodes; node id = position in graph_t::vertex_list()
 *   k[v]         vertex_t::selectable_num
 *   in_ptr/in_idx  CSR of parents, ascending node id (the order graph_t::in_edges
 *                produces, graph.hpp:389-402)
 *   cpt_off/cpt  row-major CPT of v: row = mixed radix over parents, FIRST parent
 *                most significant (all_combination_pattern order,
 *                belief_propagation.hpp:269-295), own state fastest
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n;
    const int32_t *k;
    const int32_t *in_ptr, *in_idx;
    const int64_t *cpt_off;
    const double *cpt;
    /* derived */
    int64_t E;
    int32_t *out_ptr, *out_idx, *out_edge; /* children ascending; CSR edge id of (v->child) */
    int64_t *node_off;                     /* prefix sum of k[v] */
    int64_t *msg_off;                      /* per CSR edge e=(p->v): prefix sum of k[p] */
    uint8_t *frozen;                       /* evidence marker (preconditional_node_) */
    double *pi, *lam, *pim, *lkm;          /* current state */
    double *npi, *nlam, *npim, *nlkm;      /* next state */
} oracle_bp;

/* std::max(a, b) as libstdc++ defines it: (a < b) ? b : a  -- drops a NaN b. */
static double std_max(double a, double b) { return (a < b) ? b : a; }

/* belief_propagation.hpp:298-311 normalize(): plain left-to-right sum from 0,
 * divide every element, no zero guard. */
static void normalize(double *t, int len) {
    double sum = 0;
    for (int j = 0; j < len; ++j) sum += t[j];
    for (int j = 0; j < len; ++j) t[j] /= sum;
}

static void oracle_free(oracle_bp *o) {
    free(o->out_ptr); free(o->out_idx); free(o->out_edge); free(o->node_off);
    free(o->msg_off); free(o->frozen);
    free(o->pi); free(o->lam); free(o->pim); free(o->lkm);
    free(o->npi); free(o->nlam); free(o->npim); free(o->nlkm);
}

static int oracle_build(oracle_bp *o) {
    int n = o->n;
    o->E = o->in_ptr[n];
    o->out_ptr = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    o->out_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(o->E ? o->E : 1));
    o->out_edge = (int32_t *)malloc(sizeof(int32_t) * (size_t)(o->E ? o->E : 1));
    o->node_off = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
    o->msg_off = (int64_t *)malloc(sizeof(int64_t) * ((size_t)o->E + 1));
    o->frozen = (uint8_t *)calloc((size_t)n + 1, 1);
    if (!o->out_ptr || !o->out_idx || !o->out_edge || !o->node_off || !o->msg_off || !o->frozen) return -1;
    o->node_off[0] = 0;
    for (int v = 0; v < n; ++v) o->node_off[v + 1] = o->node_off[v] + o->k[v];
    o->msg_off[0] = 0;
    for (int64_t e = 0; e < o->E; ++e) o->msg_off[e + 1] = o->msg_off[e] + o->k[o->in_idx[e]];
    /* children ascending: scanning v ascending and appending gives ascending children,
     * the order graph_t::out_edges produces (graph.hpp:362-375). */
    for (int64_t e = 0; e < o->E; ++e) o->out_ptr[o->in_idx[e] + 1]++;
    for (int v = 0; v < n; ++v) o->out_ptr[v + 1] += o->out_ptr[v];
    int32_t *fill = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    if (!fill) return -1;
    for (int v = 0; v < n; ++v)
        for (int32_t e = o->in_ptr[v]; e < o->in_ptr[v + 1]; ++e) {
            int p = o->in_idx[e];
            int32_t pos = o->out_ptr[p] + fill[p]++;
            o->out_idx[pos] = v;
            o->out_edge[pos] = e;
        }
    free(fill);
    size_t nv = (size_t)o->node_off[n] + 1, nm = (size_t)o->msg_off[o->E] + 1;
    o->pi = (double *)malloc(sizeof(double) * nv);  o->lam = (double *)malloc(sizeof(double) * nv);
    o->npi = (double *)malloc(sizeof(double) * nv); o->nlam = (double *)malloc(sizeof(double) * nv);
    o->pim = (double *)malloc(sizeof(double) * nm); o->lkm = (double *)malloc(sizeof(double) * nm);
    o->npim = (double *)malloc(sizeof(double) * nm); o->nlkm = (double *)malloc(sizeof(double) * nm);
    if (!o->pi || !o->lam || !o->npi || !o->nlam || !o->pim || !o->lkm || !o->npim || !o->nlkm) return -1;
    return 0;
}

/* belief_propagation.hpp:33-73: initial state + evidence. */
static void oracle_init(oracle_bp *o, int ne, const int32_t *ev_node, const int32_t *ev_off,
                        const double *ev_val) {
    int n = o->n;
    for (int64_t i = 0; i < o->node_off[n]; ++i) o->pi[i] = o->lam[i] = 1.0;   /* :38,:41 */
    for (int64_t i = 0; i < o->msg_off[o->E]; ++i) o->pim[i] = o->lkm[i] = 1.0; /* :44-55 */
    for (int v = 0; v < n; ++v)
        if (o->in_ptr[v] == o->in_ptr[v + 1]) /* root: pi = cpt[{}] un-normalised, :58-64 */
            memcpy(o->pi + o->node_off[v], o->cpt + o->cpt_off[v], sizeof(double) * (size_t)o->k[v]);
    memset(o->frozen, 0, (size_t)n);
    for (int j = 0; j < ne; ++j) { /* :68-73, both pi and lambda take the evidence vector */
        int v = ev_node[j];
        o->frozen[v] = 1;
        for (int i = 0; i < o->k[v]; ++i)
            o->pi[o->node_off[v] + i] = o->lam[o->node_off[v] + i] = ev_val[ev_off[j] + i];
    }
}

/* belief_propagation.hpp:202-218 calculate_pi_i(from = child v, target = parent p):
 * pi(p) times the lambda-messages of p's OTHER children in ascending order, normalised. */
static void calc_pi_i(oracle_bp *o, int v, int32_t e) {
    int p = o->in_idx[e];
    int kp = o->k[p];
    double *out = o->npim + o->msg_off[e];
    for (int i = 0; i < kp; ++i) out[i] = o->pi[o->node_off[p] + i];
    for (int i = 0; i < kp; ++i)
        for (int32_t q = o->out_ptr[p]; q < o->out_ptr[p + 1]; ++q) {
            if (o->out_idx[q] == v) continue;
            out[i] *= o->lkm[o->msg_off[o->out_edge[q]] + i];
        }
    normalize(out, kp);
}

/* belief_propagation.hpp:240-266 calculate_lambda_k(from = child v, target = parent at
 * in-edge slot jt): child state i outer, parent assignment inner (first parent slowest);
 * value = lambda(v)[i] * cpt[cond][i], times the pi-messages of the OTHER parents
 * (the reference iterates an unordered_map here, :253; ascending parent order is used). */
static void calc_lambda_k(oracle_bp *o, int v, int jt) {
    int32_t e0 = o->in_ptr[v];
    int m = o->in_ptr[v + 1] - e0;
    int kv = o->k[v];
    int kt = o->k[o->in_idx[e0 + jt]];
    double *out = o->nlkm + o->msg_off[e0 + jt];
    const double *cpt = o->cpt + o->cpt_off[v];
    int64_t rows = 1;
    int kp[64];
    for (int j = 0; j < m; ++j) { kp[j] = o->k[o->in_idx[e0 + j]]; rows *= kp[j]; }
    for (int i = 0; i < kt; ++i) out[i] = 0.0;
    int st[64];
    for (int i = 0; i < kv; ++i) {
        double times = o->lam[o->node_off[v] + i];
        for (int j = 0; j < m; ++j) st[j] = 0;
        for (int64_t r = 0; r < rows; ++r) {
            double value = times * cpt[r * kv + i];
            for (int j = 0; j < m; ++j)
                if (j != jt) value *= o->pim[o->msg_off[e0 + j] + st[j]];
            out[st[jt]] += value;
            for (int j = m - 1; j >= 0; --j) { /* odometer, last parent fastest */
                if (++st[j] < kp[j]) break;
                st[j] = 0;
            }
        }
    }
    normalize(out, kt);
}

/* belief_propagation.hpp:174-200 calculate_pi: assignment outer, own state inner;
 * value = cpt[cond][i] times pi-messages in ascending parent order. */
static void calc_pi(oracle_bp *o, int v) {
    int32_t e0 = o->in_ptr[v];
    int m = o->in_ptr[v + 1] - e0;
    int kv = o->k[v];
    double *out = o->npi + o->node_off[v];
    if (o->frozen[v]) { /* :177 evidence nodes are never updated */
        for (int i = 0; i < kv; ++i) out[i] = o->pi[o->node_off[v] + i];
        return;
    }
    const double *cpt = o->cpt + o->cpt_off[v];
    int64_t rows = 1;
    int kp[64], st[64];
    for (int j = 0; j < m; ++j) { kp[j] = o->k[o->in_idx[e0 + j]]; rows *= kp[j]; st[j] = 0; }
    for (int i = 0; i < kv; ++i) out[i] = 0.0;
    for (int64_t r = 0; r < rows; ++r) {
        for (int i = 0; i < kv; ++i) {
            double value = cpt[r * kv + i];
            for (int j = 0; j < m; ++j) value *= o->pim[o->msg_off[e0 + j] + st[j]];
            out[i] += value;
        }
        for (int j = m - 1; j >= 0; --j) {
            if (++st[j] < kp[j]) break;
            st[j] = 0;
        }
    }
    normalize(out, kv);
}

/* belief_propagation.hpp:220-238 calculate_lambda: product of children's lambda-messages
 * from 1.0 in ascending child order, normalised. */
static void calc_lambda(oracle_bp *o, int v) {
    int kv = o->k[v];
    double *out = o->nlam + o->node_off[v];
    if (o->frozen[v]) {
        for (int i = 0; i < kv; ++i) out[i] = o->lam[o->node_off[v] + i];
        return;
    }
    for (int i = 0; i < kv; ++i) {
        out[i] = 1.0;
        for (int32_t q = o->out_ptr[v]; q < o->out_ptr[v + 1]; ++q)
            out[i] *= o->lkm[o->msg_off[o->out_edge[q]] + i];
    }
    normalize(out, kv);
}

/* One Jacobi sweep, belief_propagation.hpp:75-147. Returns maximum_difference. */
static double oracle_sweep(oracle_bp *o, int threads) {
    int n = o->n;
    (void)threads;
    /* Jacobi: every output is written by exactly one iteration and only OLD state is
     * read, so the loops over v may run on several host threads with identical results. */
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (int v = 0; v < n; ++v) { /* message phase :78-88 */
        for (int32_t e = o->in_ptr[v]; e < o->in_ptr[v + 1]; ++e) calc_pi_i(o, v, e);
        int m = o->in_ptr[v + 1] - o->in_ptr[v];
        for (int j = 0; j < m; ++j) calc_lambda_k(o, v, j);
    }
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (int v = 0; v < n; ++v) { /* node phase :91-101 */
        calc_pi(o, v);
        calc_lambda(o, v);
    }
    double md = DBL_MIN; /* :105 std::numeric_limits<double>::min() */
    for (int64_t i = 0; i < o->msg_off[o->E]; ++i) { /* :106-131, messages only */
        md = std_max(md, fabs(o->npim[i] - o->pim[i]));
        md = std_max(md, fabs(o->nlkm[i] - o->lkm[i]));
    }
    double *t; /* commit :135-143 as a buffer swap */
    t = o->pi; o->pi = o->npi; o->npi = t;
    t = o->lam; o->lam = o->nlam; o->nlam = t;
    t = o->pim; o->pim = o->npim; o->npim = t;
    t = o->lkm; o->lkm = o->nlkm; o->nlkm = t;
    return md;
}

/*
 * Run BP to convergence.  max_sweeps == 0 means unbounded like the reference.
 * res_hist (optional, capacity res_cap) receives the per-sweep maximum_difference.
 * msg_dump (optional): after the final sweep receives pi-messages then lambda-messages
 * (each sum_e k[parent(e)] doubles, CSR edge order).
 * Returns 0, or -1 on allocation failure, -2 on bad arguments.
 */
int oracle_bp_run_mt(int n, const int32_t *k, const int32_t *in_ptr, const int32_t *in_idx,
                     const int64_t *cpt_off, const double *cpt, int ne, const int32_t *ev_node,
                     const int32_t *ev_off, const double *ev_val, double eps, int max_sweeps,
                     double *beliefs_out, int *sweeps_out, double *res_hist, int res_cap,
                     double *msg_dump, int threads) {
    if (n < 0) return -2;
    oracle_bp o;
    memset(&o, 0, sizeof o);
    o.n = n; o.k = k; o.in_ptr = in_ptr; o.in_idx = in_idx; o.cpt_off = cpt_off; o.cpt = cpt;
    for (int v = 0; v < n; ++v)
        if (in_ptr[v + 1] - in_ptr[v] > 64) return -2;
    if (oracle_build(&o) != 0) { oracle_free(&o); return -1; }
    oracle_init(&o, ne, ev_node, ev_off, ev_val);
    int sweeps = 0;
    while (1) {
        double md = oracle_sweep(&o, threads);
        if (res_hist && sweeps < res_cap) res_hist[sweeps] = md;
        ++sweeps;
        if (md < eps) break; /* :147 strict < */
        if (max_sweeps > 0 && sweeps >= max_sweeps) break;
    }
    for (int v = 0; v < n; ++v) { /* :151-158 belief = normalize(pi % lambda) */
        double *b = beliefs_out + o.node_off[v];
        for (int i = 0; i < k[v]; ++i) b[i] = o.pi[o.node_off[v] + i] * o.lam[o.node_off[v] + i];
        normalize(b, k[v]);
    }
    if (msg_dump) {
        int64_t nm = o.msg_off[o.E];
        memcpy(msg_dump, o.pim, sizeof(double) * (size_t)nm);
        memcpy(msg_dump + nm, o.lkm, sizeof(double) * (size_t)nm);
    }
    if (sweeps_out) *sweeps_out = sweeps;
    oracle_free(&o);
    return 0;
}

/* Single-threaded entry (the reference has no threading). */
int oracle_bp_run(int n, const int32_t *k, const int32_t *in_ptr, const int32_t *in_idx,
                  const int64_t *cpt_off, const double *cpt, int ne, const int32_t *ev_node,
                  const int32_t *ev_off, const double *ev_val, double eps, int max_sweeps,
                  double *beliefs_out, int *sweeps_out, double *res_hist, int res_cap,
                  double *msg_dump) {
    return oracle_bp_run_mt(n, k, in_ptr, in_idx, cpt_off, cpt, ne, ev_node, ev_off, ev_val, eps,
                            max_sweeps, beliefs_out, sweeps_out, res_hist, res_cap, msg_dump, 1);
}

/* ---------------------------------------------------------------------------------------------
 * Stateful API for the multi-process tests (tests/test_dist_cpu.py): the same sweep, but a rank
 * computes only what it OWNS under an edge-cut partition -- the pi-message of edge p->v when it
 * owns p, the lambda-message when it owns v, pi(v)/lambda(v) when it owns v -- and leaves the
 * rest of the next-state arrays untouched for the exchange to fill in.  Jacobi: the union over
 * ranks equals oracle_sweep exactly.
 * ------------------------------------------------------------------------------------------- */
void *oracle_bp_open(int n, const int32_t *k, const int32_t *in_ptr, const int32_t *in_idx,
                     const int64_t *cpt_off, const double *cpt) {
    oracle_bp *o = (oracle_bp *)calloc(1, sizeof(oracle_bp));
    if (!o) return NULL;
    o->n = n; o->k = k; o->in_ptr = in_ptr; o->in_idx = in_idx; o->cpt_off = cpt_off; o->cpt = cpt;
    if (oracle_build(o) != 0) { oracle_free(o); free(o); return NULL; }
    return o;
}

void oracle_bp_close(void *h) {
    if (!h) return;
    oracle_free((oracle_bp *)h);
    free(h);
}

void oracle_bp_reset(void *h, int ne, const int32_t *ev_node, const int32_t *ev_off, const double *ev_val) {
    oracle_init((oracle_bp *)h, ne, ev_node, ev_off, ev_val);
}

/* which: 0 pi, 1 lambda, 2 pi-messages, 3 lambda-messages, 4..7 the same for the NEXT state */
double *oracle_bp_array(void *h, int which) {
    oracle_bp *o = (oracle_bp *)h;
    double *a[8] = {o->pi, o->lam, o->pim, o->lkm, o->npi, o->nlam, o->npim, o->nlkm};
    return (which >= 0 && which < 8) ? a[which] : NULL;
}

/* Computes this rank's share of the next state; returns its share of maximum_difference. */
double oracle_bp_sweep_owned(void *h, const int32_t *owner, int rank) {
    oracle_bp *o = (oracle_bp *)h;
    double md = DBL_MIN;
    for (int v = 0; v < o->n; ++v) {
        int m = o->in_ptr[v + 1] - o->in_ptr[v];
        for (int j = 0; j < m; ++j) {
            int32_t e = o->in_ptr[v] + j;
            int p = o->in_idx[e];
            if (owner[p] == rank) {
                calc_pi_i(o, v, e);
                for (int i = 0; i < o->k[p]; ++i)
                    md = std_max(md, fabs(o->npim[o->msg_off[e] + i] - o->pim[o->msg_off[e] + i]));
            }
            if (owner[v] == rank) {
                calc_lambda_k(o, v, j);
                for (int i = 0; i < o->k[p]; ++i)
                    md = std_max(md, fabs(o->nlkm[o->msg_off[e] + i] - o->lkm[o->msg_off[e] + i]));
            }
        }
        if (owner[v] == rank) { calc_pi(o, v); calc_lambda(o, v); }
    }
    return md;
}

void oracle_bp_commit(void *h) {
    oracle_bp *o = (oracle_bp *)h;
    double *t;
    t = o->pi; o->pi = o->npi; o->npi = t;
    t = o->lam; o->lam = o->nlam; o->nlam = t;
    t = o->pim; o->pim = o->npim; o->npim = t;
    t = o->lkm; o->lkm = o->nlkm; o->nlkm = t;
}

/* belief of node v from the current state (belief_propagation.hpp:151-158) */
void oracle_bp_belief(void *h, int v, double *out) {
    oracle_bp *o = (oracle_bp *)h;
    for (int i = 0; i < o->k[v]; ++i) out[i] = o->pi[o->node_off[v] + i] * o->lam[o->node_off[v] + i];
    normalize(out, o->k[v]);
}

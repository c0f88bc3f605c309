// bn_lw.cpp -- host driver of the likelihood-weighting kernel: topological order, device images,
// batching, histogram read-back.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <vector>

#include "bn_lw.hpp"

namespace bnmi {

#define LWCHK(expr)                                                            \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);           \
            return BN_ERR_HIP;                                                 \
        }                                                                      \
    } while (0)

void lw_free(LwState& s) {
    void* ptrs[] = {s.d_k, s.d_node_off, s.d_cpt, s.d_thr, s.d_thr32, s.d_thr16, s.d_steps, s.d_small_steps, s.d_parents, s.d_ev_topo, s.d_states, s.d_weights, s.d_hist};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (s.h_ev) (void)hipHostFree(s.h_ev);
    if (s.h_hist) (void)hipHostFree(s.h_hist);
    s = LwState();
}

// Kahn's algorithm with a min-heap on node id: deterministic, and the identity permutation when
// every parent precedes its children.  Any topological order samples the reference's
// distribution (its own order is a DFS from the last vertex, likelihood_weighting.hpp:162-170).
static bool topo_order(const Plan& p, std::vector<int32_t>& topo) {
    const int32_t n = p.n;
    std::vector<int32_t> indeg(n), out_ptr(n + 1, 0), out_idx(std::max<int64_t>(p.E, 1));
    for (int32_t v = 0; v < n; ++v) indeg[v] = p.in_ptr[v + 1] - p.in_ptr[v];
    for (int64_t e = 0; e < p.E; ++e) out_ptr[p.in_idx[e] + 1]++;
    for (int32_t v = 0; v < n; ++v) out_ptr[v + 1] += out_ptr[v];
    std::vector<int32_t> fill(n, 0);
    for (int32_t v = 0; v < n; ++v)
        for (int32_t e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e) {
            int32_t u = p.in_idx[e];
            out_idx[out_ptr[u] + fill[u]++] = v;
        }
    std::priority_queue<int32_t, std::vector<int32_t>, std::greater<int32_t>> ready;
    for (int32_t v = 0; v < n; ++v)
        if (indeg[v] == 0) ready.push(v);
    topo.clear();
    topo.reserve(n);
    while (!ready.empty()) {
        int32_t v = ready.top();
        ready.pop();
        topo.push_back(v);
        for (int32_t q = out_ptr[v]; q < out_ptr[v + 1]; ++q)
            if (--indeg[out_idx[q]] == 0) ready.push(out_idx[q]);
    }
    return int32_t(topo.size()) == n;
}

template <class T>
static int up(T** dst, const T* src, size_t count, hipStream_t st, std::string& err) {
    LWCHK(hipMalloc(reinterpret_cast<void**>(dst), std::max<size_t>(count, 1) * sizeof(T)));
    if (count) LWCHK(hipMemcpyAsync(*dst, src, count * sizeof(T), hipMemcpyHostToDevice, st));
    return 0;
}

static int lw_prepare(LwState& s, const Plan& p, hipStream_t st, uint64_t want_samples, std::string& err) {
    if (!s.ready) {
        if (!topo_order(p, s.topo)) {
            err = "the model has a directed cycle: likelihood weighting needs a DAG (graph.hpp:268-291)";
            return BN_ERR_ARG;
        }
        // per-position descriptors: the kernel reads them with scalar loads, one position ahead
        std::vector<LwStep> steps(size_t(p.n) + 2);  // two spare: the kernel reads two positions ahead
        std::vector<LwParent> parents;
        parents.reserve(size_t(p.E) + size_t(p.n) + 4);
        s.kmax = 0;
        s.rows24 = true;
        s.inline_parents = p.n <= (1 << 24);
        // the kernel picks a state by counting the running totals u has reached, which equals the
        // reference's interval test (:177-193) when the totals never decrease
        for (double x : p.cpt_flat)
            if (!(x >= 0.0) || x > 1.7976931348623157e308) {
                err = "likelihood weighting needs finite, non-negative CPT entries";
                return BN_ERR_ARG;
            }
        s.small = s.inline_parents && p.n < (1 << 24) - 1 && p.cpt_off[p.n] < (int64_t(1) << 32);
        s.small_pow2 = true;
        for (int32_t v = 0; v < p.n && s.small; ++v) {
            const int64_t rows = (p.cpt_off[v + 1] - p.cpt_off[v]) / p.k[v];
            if (p.in_ptr[v + 1] - p.in_ptr[v] > 4 || rows > 256 || p.k[v] > 4) s.small = false;
            if (p.k[v] & (p.k[v] - 1)) s.small_pow2 = false;
        }
        if (const char* e = getenv("BN_LW_SMALL")) s.small = s.small && atoi(e) != 0;   // (A/B: 0 = the generic kernel)
        std::vector<int64_t> row16(p.n, -1);   // first row of the node's table in thr16, nodes the kernel stages in LDS only
        uint32_t rows16 = 0;
        for (int32_t t = 0; t < p.n; ++t) {
            const int32_t v = s.topo[t];
            const int64_t coff = p.cpt_off[v];
            if ((p.cpt_off[v + 1] - coff) / p.k[v] > int64_t(0xffffffffu) || coff >= (int64_t(1) << 48)) {
                err = "CPT too large for the sampler";
                return BN_ERR_ARG;
            }
            LwStep& sd = steps[t];
            sd.coff_lo = uint32_t(uint64_t(coff));
            sd.coff_hi = uint16_t(uint64_t(coff) >> 32);
            sd.v = v;
            sd.par_off = uint32_t(parents.size());
            sd.kv = uint8_t(p.k[v]);
            sd.m = uint8_t(p.in_ptr[v + 1] - p.in_ptr[v]);
            for (int q = 0; q < 4; ++q) sd.par[q] = 0;
            for (int32_t e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e) {
                parents.push_back(LwParent{uint32_t(p.in_idx[e]), uint32_t(p.k[p.in_idx[e]])});
                const int32_t j = e - p.in_ptr[v];
                if (j < 4 && s.inline_parents) sd.par[j] = uint32_t(p.in_idx[e]) | (uint32_t(p.k[p.in_idx[e]]) << 24);
            }
            if (s.inline_parents && sd.m <= 4 && (p.cpt_off[v + 1] - coff) / p.k[v] <= 256) {
                bool pow2 = true;
                for (int32_t e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e) pow2 = pow2 && (p.k[p.in_idx[e]] & (p.k[p.in_idx[e]] - 1)) == 0;
                sd.m |= uint8_t(kLwStepPacked | (pow2 ? kLwStepPow2 : 0));
                if (p.k[v] <= 4) {
                    row16[v] = rows16;
                    sd.par_off = rows16;   // (<= 4 parents: all inline, the list offset is not looked at)
                    rows16 += uint32_t(((p.cpt_off[v + 1] - coff) / p.k[v] + 1) & ~int64_t(1));
                }
            }
            if (s.small) {   // lw_sample_small_kernel's descriptor (node numbers for now: the row stride is not known yet)
                if (s.h_small.empty()) s.h_small.assign(size_t(p.n) + 3, LwSmallStep{{uint64_t(p.n), uint64_t(p.n), uint64_t(p.n), uint64_t(p.n)}, uint64_t(p.n), {0, 0, 0, 0}, 0, 0});
                LwSmallStep& ss = s.h_small[t];
                const int32_t m = p.in_ptr[v + 1] - p.in_ptr[v];
                uint32_t shape = uint32_t(p.k[v]) | (uint32_t(m) << 4);
                for (int32_t j = 0; j < 4; ++j) {
                    const int32_t node = j < m ? p.in_idx[p.in_ptr[v] + j] : p.n;
                    const uint32_t kk = j < m ? uint32_t(p.k[node]) : 1u;
                    ss.par[j] = uint64_t(node);
                    if (j > 0) shape |= (s.small_pow2 ? uint32_t(__builtin_ctz(kk)) : kk) << (8 * j);
                }
                // the sampling kernel loads a position's parents two positions ahead: a parent that is the node of the position before (bit 7,
                // which one: bits 12-13) or of the one before that (bit 22, bits 20-21) is patched from that position's store instead
                for (int32_t j = 0; j < m; ++j) {
                    if (t > 0 && p.in_idx[p.in_ptr[v] + j] == s.topo[t - 1]) shape |= 0x80u | (uint32_t(j) << 12);
                    if (t > 1 && p.in_idx[p.in_ptr[v] + j] == s.topo[t - 2]) shape |= 0x400000u | (uint32_t(j) << 20);
                }
                ss.own = uint64_t(v);
                ss.coff = uint32_t(coff);
                ss.tab[0] = uint32_t(row16[v]);   // (host copy: first row in d_thr16 and the table's bytes; the device copy holds the descriptor)
                ss.tab[1] = uint32_t((((p.cpt_off[v + 1] - coff) / p.k[v] + 1) & ~int64_t(1)) * 8);
                ss.shape = shape;
            }
            if (parents.size() & 1) parents.push_back(LwParent{0, 1});  // pairs: 16-byte aligned loads
            s.kmax = std::max(s.kmax, p.k[v]);
            if ((p.cpt_off[v + 1] - coff) / p.k[v] >= (int64_t(1) << 24)) s.rows24 = false;
        }
        for (int q = 0; q < 4; ++q) parents.push_back(LwParent{0, 1});
        int r;
        if ((r = up(&s.d_k, p.k.data(), p.k.size(), st, err))) return r;
        if ((r = up(&s.d_node_off, p.node_off.data(), p.node_off.size(), st, err))) return r;
        if ((r = up(&s.d_cpt, p.cpt_flat.data(), p.cpt_flat.size(), st, err))) return r;
        {   // selection thresholds: the running totals of every row, added up left to right as make_random_by_weight does
            // (likelihood_weighting.hpp:177-193), as integers ceil(total * 2^53): "u >= total" for u = U * 2^-53 is "U >= threshold"
            std::vector<unsigned long long> thr(p.cpt_flat.size(), ~0ull);
            for (int32_t v = 0; v < p.n; ++v) {
                const int32_t kv = p.k[v];
                for (int64_t o = p.cpt_off[v]; o < p.cpt_off[v + 1]; o += kv) {
                    double total = 0.0;
                    for (int32_t i = 0; i + 1 < kv; ++i) {
                        total = i == 0 ? p.cpt_flat[o] : total + p.cpt_flat[o + i];
                        const double y = std::ceil(std::ldexp(total, 53));
                        thr[o + i] = y >= 18446744073709551616.0 ? ~0ull : static_cast<unsigned long long>(y);
                    }
                }
            }
            std::vector<uint32_t> top(thr.size() + 4, 0xffffffffu);   // (+ 4: a 16-byte load at the last row stays inside)
            for (size_t q = 0; q < thr.size(); ++q) top[q] = thr[q] == ~0ull ? 0xffffffffu : uint32_t(std::min<unsigned long long>(thr[q] >> 21, 0xffffffffull));
            std::vector<uint32_t> t16((size_t(rows16) + 256) * 2, 0xffffffffu);   // (+ 256 rows: the kernel copies 2 KB whatever the table's size)
            for (int32_t v = 0; v < p.n; ++v) {
                if (row16[v] < 0) continue;
                const int32_t kv = p.k[v];
                size_t q = size_t(row16[v]) * 2;
                for (int64_t o = p.cpt_off[v]; o < p.cpt_off[v + 1]; o += kv, q += 2) {
                    uint32_t e[3] = {0xffffu, 0xffffu, 0xffffu};
                    for (int32_t i = 0; i + 1 < kv; ++i) e[i] = thr[o + i] == ~0ull ? 0xffffu : uint32_t(std::min<unsigned long long>(thr[o + i] >> 37, 0xffffull));
                    t16[q] = e[0] | (e[1] << 16);
                    t16[q + 1] = e[2] | 0xffff0000u;
                }
            }
            if ((r = up(&s.d_thr16, t16.data(), t16.size(), st, err))) return r;
            if ((r = up(&s.d_thr, thr.data(), thr.size(), st, err))) return r;
            if ((r = up(&s.d_thr32, top.data(), top.size(), st, err))) return r;
            LWCHK(hipStreamSynchronize(st));  // `thr`, `top` are locals
        }
        if ((r = up(&s.d_steps, steps.data(), steps.size(), st, err))) return r;
        if ((r = up(&s.d_parents, parents.data(), parents.size(), st, err))) return r;
        LWCHK(hipStreamSynchronize(st));  // steps / parents are locals
        LWCHK(hipMalloc(reinterpret_cast<void**>(&s.d_ev_topo), (size_t(p.n) + 3) * sizeof(int32_t)));
        LWCHK(hipMemsetAsync(s.d_ev_topo, 0xff, (size_t(p.n) + 3) * sizeof(int32_t), st));
        LWCHK(hipMalloc(reinterpret_cast<void**>(&s.d_hist), std::max<size_t>(p.node_off[p.n], 1) * sizeof(double)));
        LWCHK(hipHostMalloc(reinterpret_cast<void**>(&s.h_ev), (size_t(p.n) + 1) * sizeof(int32_t), hipHostMallocDefault));
        LWCHK(hipHostMalloc(reinterpret_cast<void**>(&s.h_hist), std::max<size_t>(p.node_off[p.n], 1) * sizeof(double), hipHostMallocDefault));
        s.ready = true;
    }
    // batch: enough blocks to fill the chip, bounded so the [node][sample] state matrix stays
    // within BN_LW_STATE_GIB (default 32 of the 288 GB)
#ifndef BN_LW_STATE_GIB
#define BN_LW_STATE_GIB 32
#endif
    // (LwState::small: four samples per byte)
    const uint64_t per_byte = s.small ? 4 : 1;
    uint64_t cap = (uint64_t(BN_LW_STATE_GIB) << 30) / std::max<uint64_t>(p.n, 1) * per_byte;
    cap = std::max<uint64_t>(kLwBlockSamples, cap / kLwBlockSamples * kLwBlockSamples);
    uint64_t want = (want_samples + kLwBlockSamples - 1) / kLwBlockSamples * kLwBlockSamples;
    uint64_t batch = std::min<uint64_t>({want, cap, uint64_t(16384) * kLwBlockSamples});
    batch = std::max<uint64_t>(batch, kLwBlockSamples);
    // (Launches of whole rounds of resident waves -- 8 192 waves = 2.1 M samples on 256 CUs instead of the 13 420 waves of a 3.4 M-sample
    // batch -- were tried in round 5: 7.4 vs 7.5 ns per sample: waves of a half-empty second round simply run faster.)
    s.launch_samples = batch;   // (the state matrix may be larger, from an earlier call: its rows stay as long as they are)
    if (batch > s.batch) {
        if (s.d_states) (void)hipFree(s.d_states);
        if (s.d_weights) (void)hipFree(s.d_weights);
        s.d_states = nullptr;
        s.d_weights = nullptr;
        s.batch = 0;    // (a failure below leaves nothing that claims to be allocated: the next call starts over)
        s.stride = 0;
        // Row stride = the row's bytes + 33 x 128: never a power of two.  With rows exactly 2^21 bytes apart the histogram pass, whose 64
        // lanes read 64 consecutive rows at the same offset, ran 48 % slower per sample (4.1 vs 2.7 ns) -- every lane's line in the
        // same cache set / memory channel.
        const uint64_t stride = batch / per_byte + 33 * 128;
        LWCHK(hipMalloc(reinterpret_cast<void**>(&s.d_states), (uint64_t(p.n) + 1) * stride));
        LWCHK(hipMemsetAsync(s.d_states + uint64_t(p.n) * stride, 0, stride, st));   // row n: all zero (the generic kernel's "parent" of nodes with fewer than four inline parents)
        if (s.small) {   // the descriptors hold ADDRESSES: of rows of this state matrix, of the tables in d_thr16
            std::vector<LwSmallStep> dev(s.h_small);
            const uint64_t st_base = reinterpret_cast<uint64_t>(s.d_states), th_base = reinterpret_cast<uint64_t>(s.d_thr16);
            for (LwSmallStep& ss : dev) {
                for (uint64_t& x : ss.par) x = st_base + x * stride;
                ss.own = st_base + ss.own * stride;
                const uint64_t tb = th_base + uint64_t(ss.tab[0]) * 8;
                const uint32_t bytes = ss.tab[1];
                ss.tab[0] = uint32_t(tb);
                ss.tab[1] = uint32_t(tb >> 32) & 0xffffu;   // (stride 0)
                ss.tab[2] = bytes;
                ss.tab[3] = 0x00020000u;                    // raw buffer, dword data format
            }
            if (!s.d_small_steps) LWCHK(hipMalloc(reinterpret_cast<void**>(&s.d_small_steps), dev.size() * sizeof(LwSmallStep)));
            LWCHK(hipMemcpyAsync(s.d_small_steps, dev.data(), dev.size() * sizeof(LwSmallStep), hipMemcpyHostToDevice, st));
            LWCHK(hipStreamSynchronize(st));   // `dev` is a local
        }
        s.stride = stride;
        LWCHK(hipMalloc(reinterpret_cast<void**>(&s.d_weights), batch * sizeof(double)));
        s.batch = batch;
    }
    return 0;
}

int lw_run(LwState& s, const Plan& p, void* stream, int32_t ne, const int32_t* ev_node, const int32_t* ev_state,
           uint64_t sample_begin, uint64_t n_samples, uint64_t seed, double* hist_out, std::string& err) {
    hipStream_t st = (hipStream_t)stream;
    std::vector<int32_t> evs(std::max(p.n, 1), -1);
    for (int32_t j = 0; j < ne; ++j) {
        int32_t v = ev_node[j];
        if (v < 0 || v >= p.n) { err = "evidence node out of range"; return BN_ERR_ARG; }
        // the reference would throw std::out_of_range from .at() (likelihood_weighting.hpp:151)
        if (ev_state[j] < 0 || ev_state[j] >= p.k[v]) { err = "evidence state out of range"; return BN_ERR_ARG; }
        if (evs[v] >= 0) { err = "evidence node listed twice"; return BN_ERR_ARG; }
        evs[v] = ev_state[j];
    }
    int r = lw_prepare(s, p, st, n_samples, err);
    if (r) return r;
    const size_t hist_n = size_t(p.node_off[p.n]);
    // one synchronisation per call: the evidence goes up from page-locked staging (no kernel of an earlier call is in flight: every
    // entry point of the sampler family returns synchronised), the histogram comes down into page-locked memory
    for (int32_t t = 0; t < p.n; ++t) s.h_ev[t] = evs[s.topo[t]];
    LWCHK(hipMemcpyAsync(s.d_ev_topo, s.h_ev, sizeof(int32_t) * p.n, hipMemcpyHostToDevice, st));
    LWCHK(hipMemsetAsync(s.d_hist, 0, std::max<size_t>(hist_n, 1) * sizeof(double), st));
    uint64_t done = 0;
    while (done < n_samples) {
        const uint64_t cnt = std::min<uint64_t>(s.launch_samples, n_samples - done);
        LwArgs a{p.n, s.kmax, s.rows24, s.inline_parents, s.small, s.small_pow2, s.d_steps, s.d_small_steps, s.d_parents, s.d_ev_topo, s.d_k, s.d_node_off, s.d_cpt, s.d_thr, s.d_thr32, s.d_thr16,
                 s.d_states, s.d_weights, s.d_hist, s.stride, s.small, sample_begin + done, cnt, seed, 0};
        const int blocks = int((cnt + kLwBlockSamples - 1) / kLwBlockSamples);
        if (launch_lw_sample(a, blocks, st) || launch_lw_hist(a, blocks, st)) { err = "lw kernel launch failed"; return BN_ERR_HIP; }
        s.last_batch_samples = cnt;
        done += cnt;
    }
    if (hist_out) LWCHK(hipMemcpyAsync(s.h_hist, s.d_hist, hist_n * sizeof(double), hipMemcpyDeviceToHost, st));
    LWCHK(hipStreamSynchronize(st));
    if (hist_out) std::memcpy(hist_out, s.h_hist, hist_n * sizeof(double));
    return 0;
}

int rs_run(LwState& s, const Plan& p, void* stream, int32_t ne, const int32_t* ev_node, const int32_t* ev_state,
           uint64_t sample_begin, uint64_t n_accept, uint64_t max_draw, uint64_t seed, double* counts_out,
           uint64_t* drawn_out, uint64_t* accepted_out, std::string& err) {
    hipStream_t st = (hipStream_t)stream;
    std::vector<int32_t> evs(std::max(p.n, 1), -1);
    for (int32_t j = 0; j < ne; ++j) {
        int32_t v = ev_node[j];
        if (v < 0 || v >= p.n) { err = "condition node out of range"; return BN_ERR_ARG; }
        if (ev_state[j] < 0 || ev_state[j] >= p.k[v]) { err = "condition state out of range"; return BN_ERR_ARG; }
        evs[v] = ev_state[j];  // the reference checks every listed pair; a repeated node just repeats the test
    }
    // a few times the wanted count per round: acceptance is usually well below 1
    int r = lw_prepare(s, p, st, std::min<uint64_t>(std::max<uint64_t>(4 * n_accept, kLwBlockSamples), max_draw), err);
    if (r) return r;
    const size_t hist_n = size_t(p.node_off[p.n]);
    std::vector<int32_t> evt(std::max(p.n, 1));
    for (int32_t t = 0; t < p.n; ++t) evt[t] = evs[s.topo[t]];
    LWCHK(hipMemcpyAsync(s.d_ev_topo, evt.data(), sizeof(int32_t) * p.n, hipMemcpyHostToDevice, st));
    LWCHK(hipMemsetAsync(s.d_hist, 0, std::max<size_t>(hist_n, 1) * sizeof(double), st));
    std::vector<double> w(s.batch);
    uint64_t drawn = 0, accepted = 0;
    while (accepted < n_accept && drawn < max_draw) {
        const uint64_t cnt = std::min<uint64_t>(s.launch_samples, max_draw - drawn);
        LwArgs a{p.n, s.kmax, s.rows24, s.inline_parents, s.small, s.small_pow2, s.d_steps, s.d_small_steps, s.d_parents, s.d_ev_topo, s.d_k, s.d_node_off, s.d_cpt, s.d_thr, s.d_thr32, s.d_thr16,
                 s.d_states, s.d_weights, s.d_hist, s.stride, s.small, sample_begin + drawn, cnt, seed, 1};
        const int blocks = int((cnt + kLwBlockSamples - 1) / kLwBlockSamples);
        if (launch_lw_sample(a, blocks, st)) { err = "sampling kernel launch failed"; return BN_ERR_HIP; }
        LWCHK(hipMemcpyAsync(w.data(), s.d_weights, cnt * sizeof(double), hipMemcpyDeviceToHost, st));
        LWCHK(hipStreamSynchronize(st));
        // samples count in index order until n_accept of them were accepted (rejection_sampling.hpp:93-111)
        uint64_t use = 0;
        while (use < cnt && accepted < n_accept) accepted += (w[use++] != 0.0);
        a.n_valid = use;
        if (launch_lw_hist(a, blocks, st)) { err = "histogram kernel launch failed"; return BN_ERR_HIP; }
        s.last_batch_samples = cnt;
        drawn += use;
    }
    LWCHK(hipMemcpyAsync(counts_out, s.d_hist, hist_n * sizeof(double), hipMemcpyDeviceToHost, st));
    LWCHK(hipStreamSynchronize(st));
    if (drawn_out) *drawn_out = drawn;
    if (accepted_out) *accepted_out = accepted;
    return 0;
}

int lw_states(LwState& s, const Plan& p, void* stream, uint64_t n, uint8_t* states_out, double* weights_out,
              std::string& err) {
    hipStream_t st = (hipStream_t)stream;
    if (!s.ready || s.last_batch_samples == 0) { err = "no likelihood-weighting run yet"; return BN_ERR_STATE; }
    if (n > s.last_batch_samples) { err = "more samples requested than the last batch holds"; return BN_ERR_ARG; }
    if (states_out && n > 0) {   // transposed on the device (sample-major), then one copy
        uint8_t* d_t = nullptr;
        LWCHK(hipMalloc(reinterpret_cast<void**>(&d_t), n * uint64_t(p.n)));
        int rc = launch_lw_transpose(s.d_states, d_t, p.n, s.stride, s.small, n, st);
        hipError_t ce = rc ? hipSuccess : hipMemcpyAsync(states_out, d_t, n * uint64_t(p.n), hipMemcpyDeviceToHost, st);
        if (!rc && ce == hipSuccess) ce = hipStreamSynchronize(st);
        (void)hipFree(d_t);
        if (rc) { err = "transpose kernel launch failed"; return BN_ERR_HIP; }
        if (ce != hipSuccess) { err = std::string("copying the sampled states: ") + hipGetErrorString(ce); return BN_ERR_HIP; }
    }
    if (weights_out) {
        LWCHK(hipMemcpyAsync(weights_out, s.d_weights, n * sizeof(double), hipMemcpyDeviceToHost, st));
        LWCHK(hipStreamSynchronize(st));
    }
    return 0;
}

}  // namespace bnmi

// bn_small_plan.cpp -- host plan of the one-workgroup path for small networks (bn_small.hpp, bn_small.hip).
//
// Encodings (all indices are element indices into the LDS arrays of bn_small.hip, < 65 536):
//   SmallEntry.x  = element of lambda(v) the entry multiplies by | place of its pi(v) term << 16
//   SmallEntry.y  = first parent term in `term` | number of parents << 16 | valid << 24
//   term[t]       = element of parent j's pi-message the entry's assignment selects | place of the term of the
//                   lambda-message to parent j << 16
//   accumulator slot: x = first staged term of the element's run | padded run length << 16 (a multiple of 4, the same for
//                         every lane of the wave)
//                     y = output element | arity of the vector << 16 | lane of the vector's first element << 24
//                     z = kind (0 none, 1 pi(v), 2 lambda-message)
//   product slot:     x = first child-list entry | number of children << 16
//                     y = as above
//                     z = kind (3 lambda(v), 4 pi-message) | ordinal of the child the message goes to << 8 (0xffff: none)
//                     w = element of pi(u) the message starts from
//
// Staging layout.  Every accumulator element owns a contiguous run: its terms in the reference's summation order, padded
// with zeros (written once, never touched again: x + 0.0 == x for the partial sums that occur) to the longest run of
// its wave's row rounded up to 4 -- so every lane of a wave adds the same number of terms with the same instructions,
// reads with immediate offsets, no selects.  The runs of a vector's elements start an ODD number of words apart: the
// lanes of a wave then hit different LDS banks (runs a power of two apart cost 16- to 32-way conflicts on every read
// of the dependent chain).
#include "bn_small.hpp"

#include <algorithm>
#include <cstdlib>

namespace bnmi {

namespace {

struct Vec {           // one output vector = k adjacent lanes of one wavefront
    int cost;          // terms (accumulator) / children (product) per element: orders the packing
    int k;
    SmallSlot first;   // slot of element 0; element i adds `step` to x / y / w
    uint32_t step_x, step_y, step_w;
    int node, which;   // accumulators: the node and -1 (pi(v)) or the in-edge slot j (lambda-message)
    int row, lane0;    // where the packing put it
};

// Vectors, most expensive first, into rows of 64 lanes (no vector straddles a row).  `alike`: a row also ends where the
// cost falls below 3/4 of its first vector's -- the lanes of a wave run in lockstep to the longest run among them --
// unless that needs more than `max_rows` rows.  Returns the number of rows.
int pack_rows(std::vector<Vec>& vecs, bool alike, int max_rows) {
    std::stable_sort(vecs.begin(), vecs.end(), [](const Vec& a, const Vec& b) { return a.cost > b.cost; });
    int nrows = 0;
    for (int pass = alike ? 0 : 1; pass < 2; ++pass) {
        nrows = 0;
        int used = kWave, first_cost = 0;
        for (Vec& v : vecs) {
            if (used + v.k > kWave || (pass == 0 && 4 * v.cost < 3 * first_cost)) {
                ++nrows;
                used = 0;
                first_cost = v.cost;
            }
            v.row = nrows - 1;
            v.lane0 = used;
            used += v.k;
        }
        if (nrows <= max_rows) break;
    }
    return nrows;
}

}  // namespace

// The plan of the items of nodes [v0, v1).  part = false: the whole network in one workgroup, state in LDS (bn_small.hip).
// part = true: one workgroup's share of a network that is spread over several (bn_mid.hip): message and node-vector indices stay
// GLOBAL (the state lives in memory), staging places, parent terms and child lists are the workgroup's own.
static void build_plan_of_range(const Plan& p, int v0, int v1, bool part, SmallPlan& sp) {
    sp = SmallPlan();
    auto no = [&](const char* why) { sp.ok = false; sp.why = why; };
    if (p.nranks != 1) return no("sharded");
    const int n = p.n;
    const int64_t E = p.E;
    if (n <= 0 || v1 <= v0) return no("empty");
    const int64_t N = p.node_off[n], M = p.msg_off[E], S = p.cpt_off[v1] - p.cpt_off[v0];
    if (n > 60000 || N > 60000 || M > 60000 || S > int64_t(kSmallMaxRounds) * kSmallMaxWaves * kWave) return no("too large");
    std::vector<int> m(n), rows(n);
    int mmax = 0;
    int64_t TT = 0;
    for (int v = v0; v < v1; ++v) {
        m[v] = p.in_ptr[v + 1] - p.in_ptr[v];
        if (m[v] > kSmallMaxParents) return no("a node has more than 8 parents");
        if (p.k[v] > kWave) return no("arity above 64");
        mmax = std::max(mmax, m[v]);
        int64_t r = 1;
        for (int j = 0; j < m[v]; ++j) r *= p.k[p.in_idx[p.in_ptr[v] + j]];
        if (r * p.k[v] != p.cpt_off[v + 1] - p.cpt_off[v]) return no("CPT size mismatch");
        if (r * p.k[v] > 30000) return no("too large");
        rows[v] = int(r);
        TT += (p.cpt_off[v + 1] - p.cpt_off[v]) * m[v];
    }
    if (TT > 65535) return no("too many staged terms");
    // children ascending (the order graph_t::out_edges produces): CSR over parents
    std::vector<int> cptr(n + 1, 0);
    for (int64_t e = 0; e < E; ++e) cptr[p.in_idx[e] + 1]++;
    for (int v = 0; v < n; ++v) cptr[v + 1] += cptr[v];
    std::vector<int> cedge(std::max<int64_t>(E, 1)), fill(n, 0);
    for (int v = 0; v < n; ++v)
        for (int e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e) cedge[cptr[p.in_idx[e]] + fill[p.in_idx[e]]++] = e;
    // the child lists of the range's nodes, one after another (cl0 = where the first one starts in the global order)
    const int cl0 = cptr[v0], ncl = cptr[v1] - cptr[v0];
    sp.clist.resize(std::max(ncl, 1), 0);
    for (int q = 0; q < ncl; ++q) sp.clist[q] = uint16_t(p.msg_off[cedge[cl0 + q]]);
    for (int v = v0; v < v1; ++v)
        if (cptr[v + 1] - cptr[v] > 60000) return no("too many children");
    if (ncl > 65000) return no("too large");

    // ---- accumulator and product vectors
    std::vector<Vec> bv, cv;
    for (int v = v0; v < v1; ++v) {
        const int kv = p.k[v], Sv = kv * rows[v];
        // pi(v): element i sums `rows` terms (the run's place is filled in once the rows are known)
        bv.push_back(Vec{rows[v], kv, SmallSlot{0u, uint32_t(p.node_off[v]) | uint32_t(kv) << 16, 1u, 0u}, 0u, 1u, 0u, v, -1, 0, 0});
        for (int j = 0; j < m[v]; ++j) {
            const int e = p.in_ptr[v] + j, kp = p.k[p.in_idx[e]];
            // kv * rows / kp terms per element of the lambda-message to parent j
            bv.push_back(Vec{Sv / kp, kp, SmallSlot{0u, uint32_t(p.msg_off[e]) | uint32_t(kp) << 16, 2u, 0u}, 0u, 1u, 0u, v, j, 0, 0});
        }
        const int deg = cptr[v + 1] - cptr[v];
        // lambda(v): product over all children
        cv.push_back(Vec{deg, kv, SmallSlot{uint32_t(cptr[v] - cl0) | uint32_t(deg) << 16, uint32_t(p.node_off[v]) | uint32_t(kv) << 16,
                                            3u | 0xffffu << 8, 0u},
                         0u, 1u, 0u, v, -1, 0, 0});
        for (int x = 0; x < deg; ++x) {  // pi-message to child x: pi(v) times the OTHER children's lambda-messages
            const int e = cedge[cptr[v] + x];
            cv.push_back(Vec{deg, kv, SmallSlot{uint32_t(cptr[v] - cl0) | uint32_t(deg) << 16, uint32_t(p.msg_off[e]) | uint32_t(kv) << 16,
                                                4u | uint32_t(x) << 8, uint32_t(p.node_off[v])},
                             0u, 1u, 1u, v, x, 0, 0});
        }
    }
    int nb = pack_rows(bv, true, kSmallPreferredWaves);
    const int nc = pack_rows(cv, false, 0);
    const int ne_rows = int((S + kWave - 1) / kWave);
    // As many waves as the largest kind needs for one round, but 12 rather than 16: three waves per SIMD instead of four
    // measured faster even where the entry items then take a second round (27-node network: 58.7 vs 61.5 us per query; ALARM-
    // shaped: 49 vs 51) -- unless the entry items already take three or more rounds at 16 waves and 12 would add another
    // (48-node network: 73 vs 65; 8 x 8 grid, k = 4: not eligible at 12).  A workgroup's share of a network spread over several
    // (part): 16 waves also where they take the entry items from three rounds to two (10 k-node mixed-arity network, 213 workgroups:
    // 9.0 -> 8.4 us per sweep; the smaller networks' parts need one round either way).
    int waves = std::max(1, std::min(kSmallPreferredWaves, std::max(ne_rows, std::max(nb, nc))));
    {
        const int re12 = (ne_rows + kSmallPreferredWaves - 1) / kSmallPreferredWaves, re16 = (ne_rows + kSmallMaxWaves - 1) / kSmallMaxWaves;
        const int rb12 = (nb + kSmallPreferredWaves - 1) / kSmallPreferredWaves, rc12 = (nc + kSmallPreferredWaves - 1) / kSmallPreferredWaves;
        if ((re12 > re16 && (re16 >= 3 || (part && re12 >= 3))) || re12 > kSmallMaxRounds || rb12 > kSmallMaxRounds || rc12 > kSmallMaxRounds) {
            waves = kSmallMaxWaves;
            nb = pack_rows(bv, true, kSmallMaxWaves);  // (more rows of like runs fit one round now)
        }
    }
    if (const char* w = std::getenv("BN_SMALL_WAVES")) waves = std::max(1, std::min(kSmallMaxWaves, std::atoi(w)));  // experiments
    sp.re = (ne_rows + waves - 1) / waves;
    sp.rb = (nb + waves - 1) / waves;
    sp.rc = (nc + waves - 1) / waves;
    if (sp.re > kSmallMaxRounds || sp.rb > kSmallMaxRounds || sp.rc > kSmallMaxRounds) return no("too many work items");
    const int nt = waves * kWave;

    // ---- staging: per row the padded run length, per vector its base and the (odd) distance between its elements' runs
    std::vector<int> row_len(nb, 0), row_pad(nb, 0);
    for (const Vec& v : bv) row_len[v.row] = std::max(row_len[v.row], v.cost);
    for (int r = 0; r < nb; ++r) row_pad[r] = (row_len[r] + 3) & ~3;
    std::vector<std::vector<uint32_t>> vec_base(n), vec_stride(n);  // [node][1 + j]
    for (int v = v0; v < v1; ++v) { vec_base[v].assign(m[v] + 1, 0u); vec_stride[v].assign(m[v] + 1, 0u); }
    int64_t T = 0;
    for (Vec& v : bv) {
        const uint32_t stride = uint32_t(row_pad[v.row]) | 1u;
        if (T + int64_t(v.k) * stride > 65000) return no("too many staged terms");
        vec_base[v.node][v.which + 1] = uint32_t(T);
        vec_stride[v.node][v.which + 1] = stride;
        v.first.x = uint32_t(T) | uint32_t(row_pad[v.row]) << 16;
        v.step_x = stride;
        T += int64_t(v.k) * stride;
    }
    T += 16;  // the loads run one step ahead of the additions: up to 12 words past the last run's end
    if (!part && T < N) T = N;  // (the final beliefs are normalised in this array)

    // ---- rows -> waves.  Accumulator rows and product rows run between the same two barriers, and wave w issues on
    // SIMD w % 4: each row, most expensive first, goes to the wave whose SIMD has the least work so far (then the wave
    // with the least), so that the wave with the longest chain of dependent additions shares its SIMD with little else.
    struct RowJob { double cost; int kind, row; };
    std::vector<RowJob> jobs;
    for (int r = 0; r < nb; ++r) jobs.push_back(RowJob{12.0 + 1.0 * row_pad[r], 0, r});
    {
        std::vector<int> dmax(nc, 0);
        for (const Vec& v : cv) dmax[v.row] = std::max(dmax[v.row], v.cost);
        for (int r = 0; r < nc; ++r) jobs.push_back(RowJob{14.0 + 4.0 * ((dmax[r] + 3) & ~3), 1, r});
    }
    std::stable_sort(jobs.begin(), jobs.end(), [](const RowJob& a, const RowJob& b) { return a.cost > b.cost; });
    std::vector<double> simd_load(4, 0.0), wave_load(waves, 0.0);
    std::vector<int> used_b(waves, 0), used_c(waves, 0);
    std::vector<int> b_wave(nb, 0), b_round(nb, 0), c_wave(nc, 0), c_round(nc, 0);
    for (const RowJob& j : jobs) {
        int best = -1;
        for (int w = 0; w < waves; ++w) {
            if ((j.kind == 0 ? used_b[w] : used_c[w]) >= (j.kind == 0 ? sp.rb : sp.rc)) continue;
            if (best < 0 || simd_load[w & 3] < simd_load[best & 3] ||
                (simd_load[w & 3] == simd_load[best & 3] && wave_load[w] < wave_load[best]))
                best = w;
        }
        if (j.kind == 0) { b_wave[j.row] = best; b_round[j.row] = used_b[best]++; }
        else { c_wave[j.row] = best; c_round[j.row] = used_c[best]++; }
        simd_load[best & 3] += j.cost;
        wave_load[best] += j.cost;
    }
    auto place = [&](const std::vector<Vec>& vecs, const std::vector<int>& wave_of, const std::vector<int>& round_of, int rounds,
                     std::vector<SmallSlot>& out) {
        out.assign(size_t(rounds) * nt, SmallSlot{0, 0, 0, 0});
        for (const Vec& v : vecs)
            for (int i = 0; i < v.k; ++i) {
                SmallSlot s = v.first;
                s.x += v.step_x * uint32_t(i);
                s.y += v.step_y * uint32_t(i);
                s.w += v.step_w * uint32_t(i);
                s.y |= uint32_t(v.lane0) << 24;  // lane of element 0
                out[size_t(round_of[v.row]) * nt + wave_of[v.row] * kWave + v.lane0 + i] = s;
            }
    };
    place(bv, b_wave, b_round, sp.rb, sp.bslot);
    place(cv, c_wave, c_round, sp.rc, sp.cslot);
    // every lane of a wave adds the same number of terms: the lanes without an item carry the row's length too
    for (int r = 0; r < nb; ++r)
        for (int l = 0; l < kWave; ++l) {
            SmallSlot& s = sp.bslot[size_t(b_round[r]) * nt + b_wave[r] * kWave + l];
            if (s.z == 0) s.x = uint32_t(row_pad[r]) << 16;
        }

    // ---- entry items: nodes ordered by parent count (a wave's entries then need the same unrolled code), entries
    // of a node in table order
    std::vector<int> order(v1 - v0);
    for (int v = v0; v < v1; ++v) order[v - v0] = v;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return m[a] > m[b]; });
    sp.ent.assign(size_t(sp.re) * nt, SmallEntry{0, 0});
    sp.ent_cpt.assign(size_t(sp.re) * nt, 0.0);
    sp.term.assign(std::max<int64_t>(TT, 1), 0);
    int64_t slot = 0, tnext = 0;
    for (int v : order) {
        const int kv = p.k[v], e0 = p.in_ptr[v];
        int kp[kSmallMaxParents], digit[kSmallMaxParents];
        for (int j = 0; j < m[v]; ++j) { kp[j] = p.k[p.in_idx[e0 + j]]; digit[j] = 0; }
        for (int a = 0; a < rows[v]; ++a) {  // odometer over the parents, last parent fastest (belief_propagation.hpp:269-295)
            for (int i = 0; i < kv; ++i) {
                // pi(v)[i] adds its terms in assignment order (:174-200): term a of element i
                const uint32_t place_pi = vec_base[v][0] + uint32_t(i) * vec_stride[v][0] + uint32_t(a);
                sp.ent[slot] = SmallEntry{uint32_t(p.node_off[v] + i) | place_pi << 16, uint32_t(tnext) | uint32_t(m[v]) << 16 | 1u << 24};
                sp.ent_cpt[slot] = p.cpt_flat[p.cpt_off[v] + int64_t(a) * kv + i];
                for (int jt = 0; jt < m[v]; ++jt) {
                    // element digit[jt] of the lambda-message to parent jt adds own state outer, assignment inner (:240-266):
                    // this entry is term i * (rows / kp) + (rank of the assignment among those that share digit jt)
                    int rest = 0;
                    for (int j = 0; j < m[v]; ++j)
                        if (j != jt) rest = rest * kp[j] + digit[j];
                    const uint32_t place_lam = vec_base[v][1 + jt] + uint32_t(digit[jt]) * vec_stride[v][1 + jt] +
                                               uint32_t(i * (rows[v] / kp[jt]) + rest);
                    sp.term[tnext++] = uint32_t(p.msg_off[e0 + jt] + digit[jt]) | place_lam << 16;
                }
                ++slot;
            }
            for (int j = m[v] - 1; j >= 0; --j) {
                if (++digit[j] < kp[j]) break;
                digit[j] = 0;
            }
        }
    }
    // ---- per node-vector element: where the evidence kernel leaves the evidence, the initial pi(v)
    sp.nv_idx.assign(N, 0);
    sp.nv_slot.assign(N, 0);
    sp.npi_init.assign(N, 1.0);
    for (int v = 0; v < n && (!part || v0 == 0); ++v) {  // (global tables: built with the first part only)
        const int mv = p.in_ptr[v + 1] - p.in_ptr[v];
        const TileDesc& td = p.tiles[p.node_tile[v]];
        const int nl = p.node_nl[v];
        for (int i = 0; i < p.k[v]; ++i) {
            // element i of pi(v) in the tile's striped node block (bn_tiles.hpp vidx(0, i, npt, nl))
            sp.nv_idx[p.node_off[v] + i] = int32_t(td.node_base + int64_t(i >> 1) * (int64_t(td.npt) * 2) + nl * 2 + (i & 1));
            sp.nv_slot[p.node_off[v] + i] = p.node_slot[v];
            if (mv == 0) sp.npi_init[p.node_off[v] + i] = p.cpt_flat[p.cpt_off[v] + i];  // a root starts from its CPT row (:58-64)
        }
    }
    sp.node_off.assign(p.node_off.begin(), p.node_off.begin() + n + 1);
    sp.n = n; sp.N = int32_t(N); sp.M = int32_t(M); sp.S = int32_t(S); sp.T = int32_t(T); sp.TT = int32_t(TT);
    sp.CL = int32_t(ncl); sp.waves = waves; sp.mmax = mmax;
    sp.v0 = v0; sp.v1 = v1;
    // LDS: 4 M + 4 N + T doubles, TT words, CL halfwords, N marks, the residual words
    size_t bytes = size_t(part ? T : 4 * M + 4 * N + T) * 8 + ((size_t(std::max<int64_t>(TT, 1)) + 1) & ~size_t(1)) * 4;
    bytes += (size_t(std::max(ncl, 1)) * 2 + 7) & ~size_t(7);
    bytes += part ? size_t(kSmallMaxWaves) * kWave * 8 : (size_t(N) + 7) & ~size_t(7);  // (parts: a normalisation scratch line per wave)
    bytes += 2 * 16 * 8;  // the residual words: [iteration parity][wave]
    if (part) bytes += 64 + 16;  // (the grid barrier's flag word; the maximum it collects)
    sp.lds_bytes = bytes;
    if (bytes > size_t(kSmallLdsBytes)) return no("state does not fit the LDS");
    sp.ok = true;
}

void build_small_plan(const Plan& p, SmallPlan& sp) { build_plan_of_range(p, 0, p.n, false, sp); }

// A network too large for one workgroup, spread over up to kMidMaxParts of them (bn_mid.hip): contiguous node ranges, each
// within the per-workgroup limits of the items (LDS, rounds).  The ranges are cut where the estimated staging of a range
// reaches a target; the target grows until 32 parts hold the network.
void build_mid_plan(const Plan& p, MidPlan& mp) {
    mp = MidPlan();
    auto no = [&](const std::string& why) { mp.ok = false; mp.why = why; mp.parts.clear(); };
    if (p.nranks != 1 || p.n <= 1) return no("sharded or empty");
    const int n = p.n;
    if (p.node_off[n] > 60000 || p.msg_off[p.E] > 60000) return no("too large");
    std::vector<int64_t> est(n);  // staged terms of node v, padding included (estimate)
    int64_t total = 0;
    for (int v = 0; v < n; ++v) {
        const int m = p.in_ptr[v + 1] - p.in_ptr[v];
        if (m > kSmallMaxParents) return no("a node has more than 8 parents");
        if (p.k[v] > kWave) return no("arity above 64");
        est[v] = (p.cpt_off[v + 1] - p.cpt_off[v]) * (m + 1) * 5 / 4 + 8 * p.k[v] * (m + 1);
        total += est[v];
    }
    // Parts of >= 2 000 staged terms, about kMidPreferredParts of them where the network is large enough: fewer, larger parts mean
    // more rounds of items per thread (stamps: 1.7 us of entry items + 1.7 us of accumulator items per iteration at four rounds),
    // and with the flag barrier more parts cost little -- us per sweep at 32 / 96 preferred: 600-node network 5.7 / 5.2, 1 000 nodes
    // 6.5 / 5.5, 3 000 nodes 7.5 / 6.8, 200 nodes of 1 024-entry tables 7.6 / 6.9, smaller networks unchanged (the 2 000-term floor
    // decides there; at 1 000 terms a 16 x 16 grid goes 4.3 -> 4.9).  Larger parts only where kMidMaxParts do not hold the network.
    int64_t preferred = kMidPreferredParts;
    if (const char* pp = std::getenv("BN_MID_PARTS")) preferred = std::max(2, std::min(kMidMaxParts, std::atoi(pp)));  // experiments
    mp.est_total = total;
    constexpr int64_t kTargetMax = 12000;  // staged terms a part may hold (LDS)
    // Targets tried in turn: from total / preferred (at least 2 000, at most 9 000) UP to the largest part size -- fewer, larger parts
    // when the network needs more than kMidMaxParts of the first size -- and then DOWN from the first: the estimate is low for
    // tables whose accumulator runs are padded most (1 024-entry tables), and a part that turns out too large for a workgroup
    // asks for more, smaller parts, not fewer.
    int64_t floor_terms = 2000;
    if (const char* ff = std::getenv("BN_MID_FLOOR")) floor_terms = std::max(500, std::atoi(ff));  // experiments
    const int64_t start = std::max<int64_t>(floor_terms, std::min<int64_t>((total + preferred - 1) / preferred, 9000));
    std::vector<int64_t> targets;
    for (int64_t t = start;; t = std::min(kTargetMax, t * 4 / 3 + 1)) {  // (the growth used to step over the largest target: 9 000 -> 12 001)
        targets.push_back(t);
        if (t == kTargetMax) break;
    }
    for (int64_t t = start * 3 / 4; t >= 1500; t = t * 3 / 4) targets.push_back(t);
    for (const int64_t target : targets) {
        // balanced: as many parts as the target asks for, each about total / parts
        const int64_t nparts_want = std::max<int64_t>(2, (total + target - 1) / target);
        if (nparts_want > kMidMaxParts) continue;
        const int64_t per = (total + nparts_want - 1) / nparts_want;
        // Large parts: also cut where the part's CPT entries would go beyond two rounds of entry items at sixteen waves -- the staged-term
        // estimate balances the accumulator work, but a part of 2 070 entries takes a third round and the whole grid waits for it at the
        // barrier (10 k-node mixed-arity network: arrival skew 0.64 us median).  Without that cut where it would need more workgroups
        // than there are.
        constexpr int64_t kTwoRounds = 2 * kSmallMaxWaves * kWave;
        std::vector<int> cut;
        for (int capped = (per > 6000 ? 1 : 0); capped >= 0; --capped) {
            cut.assign(1, 0);
            int64_t acc = 0, ent = 0;
            for (int v = 0; v < n; ++v) {
                const int64_t ev = p.cpt_off[v + 1] - p.cpt_off[v];
                const bool over = acc + est[v] > per || (capped && ent + ev > kTwoRounds);
                if (acc > 0 && over && int(cut.size()) < kMidMaxParts) { cut.push_back(v); acc = 0; ent = 0; }
                acc += est[v];
                ent += ev;
            }
            if (int(cut.size()) < kMidMaxParts) break;   // (at kMidMaxParts the last part took whatever was left)
        }
        cut.push_back(n);
        std::vector<SmallPlan> parts(cut.size() - 1);
        bool all = true;
        for (size_t q = 0; q + 1 < cut.size() && all; ++q) {
            build_plan_of_range(p, cut[q], cut[q + 1], true, parts[q]);
            all = parts[q].ok;
            if (!all) mp.why = parts[q].why;
        }
        if (!all) continue;
        mp.parts.swap(parts);
        mp.ok = true;
        mp.why.clear();
        mp.waves = 0; mp.rounds = 0; mp.lds_bytes = 0;
        for (const SmallPlan& sp : mp.parts) {
            mp.waves = std::max(mp.waves, sp.waves);
            mp.rounds = std::max(mp.rounds, std::max(sp.re, std::max(sp.rb, sp.rc)));
            mp.lds_bytes = std::max(mp.lds_bytes, sp.lds_bytes);
        }
        return;
    }
    no(mp.why.empty() ? std::string("does not fit ") + std::to_string(kMidMaxParts) + " workgroups" : mp.why);
}

}  // namespace bnmi

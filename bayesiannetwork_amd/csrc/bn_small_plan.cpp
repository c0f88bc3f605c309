// bn_small_plan.cpp -- host plan of the one-workgroup path for small networks (bn_small.hpp, bn_small.hip).
//
// Encodings (all indices are element indices into the LDS arrays of bn_small.hip, < 65 536):
//   SmallEntry.x  = element of lambda(v) the entry multiplies by | place of its pi(v) term << 16
//   SmallEntry.y  = first parent term in `term` | number of parents << 16 | valid << 24
//   term[t]       = element of parent j's pi-message the entry's assignment selects | place of the term of the
//                   lambda-message to parent j << 16
//   accumulator slot: x = first staged term of the run (term r at + r * arity) | run length << 16
//                     y = output element | arity of the vector << 16 | lane of the vector's first element << 24
//                     z = kind (0 none, 1 pi(v), 2 lambda-message)
//   product slot:     x = first child-list entry | number of children << 16
//                     y = as above
//                     z = kind (3 lambda(v), 4 pi-message) | ordinal of the child the message goes to << 8 (0xffff: none)
//                     w = element of pi(u) the message starts from
#include "bn_small.hpp"

#include <algorithm>

namespace bnmi {

namespace {

struct Vec {           // one output vector = k adjacent lanes of one wavefront
    int cost;          // terms (accumulator) / children (product) per element: orders the packing
    int k;
    SmallSlot first;   // slot of element 0; element i adds `step` to x / y / w
    uint32_t step_x, step_y, step_w;
};

// Vectors, most expensive first, into rows of 64 lanes (no vector straddles a row); row r -> wave r % waves, round r / waves.
int pack_rows(std::vector<Vec>& vecs, std::vector<std::vector<SmallSlot>>& rows) {
    std::stable_sort(vecs.begin(), vecs.end(), [](const Vec& a, const Vec& b) { return a.cost > b.cost; });
    rows.clear();
    int used = kWave;
    for (const Vec& v : vecs) {
        if (used + v.k > kWave) { rows.emplace_back(kWave, SmallSlot{0, 0, 0, 0}); used = 0; }
        for (int i = 0; i < v.k; ++i) {
            SmallSlot s = v.first;
            s.x += v.step_x * uint32_t(i);
            s.y += v.step_y * uint32_t(i);
            s.w += v.step_w * uint32_t(i);
            s.y |= uint32_t(used) << 24;  // lane of element 0
            rows.back()[used + i] = s;
        }
        used += v.k;
    }
    return int(rows.size());
}

void rows_to_slots(const std::vector<std::vector<SmallSlot>>& rows, int waves, int rounds, std::vector<SmallSlot>& out) {
    const int nt = waves * kWave;
    out.assign(size_t(rounds) * nt, SmallSlot{0, 0, 0, 0});
    for (size_t r = 0; r < rows.size(); ++r) {
        const int wave = int(r % waves), round = int(r / waves);
        for (int l = 0; l < kWave; ++l) out[size_t(round) * nt + wave * kWave + l] = rows[r][l];
    }
}

}  // namespace

void build_small_plan(const Plan& p, SmallPlan& sp) {
    sp = SmallPlan();
    auto no = [&](const char* why) { sp.ok = false; sp.why = why; };
    if (p.nranks != 1) return no("sharded");
    const int n = p.n;
    const int64_t E = p.E;
    if (n <= 0) return no("empty");
    const int64_t N = p.node_off[n], M = p.msg_off[E], S = p.cpt_off[n];
    if (n > 60000 || N > 60000 || M > 60000 || S > int64_t(kSmallMaxRounds) * kSmallMaxWaves * kWave) return no("too large");
    // per node: parents, rows, staging base
    std::vector<int> m(n), rows(n);
    std::vector<int64_t> stg_base(n + 1, 0);
    int mmax = 0;
    int64_t TT = 0;
    for (int v = 0; v < n; ++v) {
        m[v] = p.in_ptr[v + 1] - p.in_ptr[v];
        if (m[v] > kSmallMaxParents) return no("a node has more than 8 parents");
        if (p.k[v] > kWave) return no("arity above 64");
        mmax = std::max(mmax, m[v]);
        int64_t r = 1;
        for (int j = 0; j < m[v]; ++j) r *= p.k[p.in_idx[p.in_ptr[v] + j]];
        if (r * p.k[v] != p.cpt_off[v + 1] - p.cpt_off[v]) return no("CPT size mismatch");
        if (r > 60000) return no("too large");
        rows[v] = int(r);
        stg_base[v + 1] = stg_base[v] + (p.cpt_off[v + 1] - p.cpt_off[v]) * (m[v] + 1);
        TT += (p.cpt_off[v + 1] - p.cpt_off[v]) * m[v];
    }
    const int64_t T = stg_base[n];
    if (T > 65535 || TT > 65535) return no("too many staged terms");
    // children ascending (the order graph_t::out_edges produces): CSR over parents
    std::vector<int> cptr(n + 1, 0);
    for (int64_t e = 0; e < E; ++e) cptr[p.in_idx[e] + 1]++;
    for (int v = 0; v < n; ++v) cptr[v + 1] += cptr[v];
    std::vector<int> cedge(std::max<int64_t>(E, 1)), fill(n, 0);
    for (int v = 0; v < n; ++v)
        for (int e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e) cedge[cptr[p.in_idx[e]] + fill[p.in_idx[e]]++] = e;
    sp.clist.resize(std::max<int64_t>(E, 1), 0);
    for (int64_t q = 0; q < E; ++q) sp.clist[q] = uint16_t(p.msg_off[cedge[q]]);
    for (int v = 0; v < n; ++v)
        if (cptr[v + 1] - cptr[v] > 60000) return no("too many children");

    // ---- accumulator and product vectors
    std::vector<Vec> bv, cv;
    for (int v = 0; v < n; ++v) {
        const int kv = p.k[v], Sv = kv * rows[v];
        const uint32_t base = uint32_t(stg_base[v]);
        // pi(v): element i sums the `rows` terms at base + r * kv + i, r = 0 .. rows - 1 (term r of the k elements of a
        // vector side by side: adjacent lanes read adjacent words -- runs laid end to end put the lanes of a wave a run
        // length apart, 16- to 32-way LDS bank conflicts on every read of the dependent chain)
        bv.push_back(Vec{rows[v], kv, SmallSlot{base | uint32_t(rows[v]) << 16, uint32_t(p.node_off[v]) | uint32_t(kv) << 16, 1u, 0u},
                         1u, 1u, 0u});
        for (int j = 0; j < m[v]; ++j) {
            const int e = p.in_ptr[v] + j, kp = p.k[p.in_idx[e]];
            const int run = Sv / kp;  // kv * rows / kp terms per element of the lambda-message to parent j
            bv.push_back(Vec{run, kp, SmallSlot{(base + uint32_t(Sv) * uint32_t(1 + j)) | uint32_t(run) << 16,
                                                uint32_t(p.msg_off[e]) | uint32_t(kp) << 16, 2u, 0u},
                             1u, 1u, 0u});
        }
        const int deg = cptr[v + 1] - cptr[v];
        // lambda(v): product over all children
        cv.push_back(Vec{deg, kv, SmallSlot{uint32_t(cptr[v]) | uint32_t(deg) << 16, uint32_t(p.node_off[v]) | uint32_t(kv) << 16,
                                            3u | 0xffffu << 8, 0u},
                         0u, 1u, 0u});
        for (int x = 0; x < deg; ++x) {  // pi-message to child x: pi(v) times the OTHER children's lambda-messages
            const int e = cedge[cptr[v] + x];
            cv.push_back(Vec{deg, kv, SmallSlot{uint32_t(cptr[v]) | uint32_t(deg) << 16, uint32_t(p.msg_off[e]) | uint32_t(kv) << 16,
                                                4u | uint32_t(x) << 8, uint32_t(p.node_off[v])},
                             0u, 1u, 1u});
        }
    }
    std::vector<std::vector<SmallSlot>> brows, crows;
    const int nb = pack_rows(bv, brows), nc = pack_rows(cv, crows);
    const int ne_rows = int((S + kWave - 1) / kWave);
    // as many waves as the largest kind needs for one round, at most 16
    int waves = std::max(1, std::min(kSmallMaxWaves, std::max(ne_rows, std::max(nb, nc))));
    sp.re = (ne_rows + waves - 1) / waves;
    sp.rb = (nb + waves - 1) / waves;
    sp.rc = (nc + waves - 1) / waves;
    if (sp.re > kSmallMaxRounds || sp.rb > kSmallMaxRounds || sp.rc > kSmallMaxRounds) return no("too many work items");
    const int nt = waves * kWave;
    rows_to_slots(brows, waves, sp.rb, sp.bslot);
    rows_to_slots(crows, waves, sp.rc, sp.cslot);

    // ---- entry items: nodes ordered by parent count (a wave's entries then need the same unrolled code), entries
    // of a node in table order
    std::vector<int> order(n);
    for (int v = 0; v < n; ++v) order[v] = v;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return m[a] > m[b]; });
    sp.ent.assign(size_t(sp.re) * nt, SmallEntry{0, 0});
    sp.ent_cpt.assign(size_t(sp.re) * nt, 0.0);
    sp.term.assign(std::max<int64_t>(TT, 1), 0);
    int64_t slot = 0, tnext = 0;
    for (int v : order) {
        const int kv = p.k[v], Sv = kv * rows[v], e0 = p.in_ptr[v];
        const uint32_t base = uint32_t(stg_base[v]);
        int kp[kSmallMaxParents], digit[kSmallMaxParents];
        for (int j = 0; j < m[v]; ++j) { kp[j] = p.k[p.in_idx[e0 + j]]; digit[j] = 0; }
        for (int a = 0; a < rows[v]; ++a) {  // odometer over the parents, last parent fastest (belief_propagation.hpp:269-295)
            for (int i = 0; i < kv; ++i) {
                const size_t at = size_t(slot / nt) * nt + size_t(slot % nt);  // round-major: slot s -> round s / nt, thread s % nt
                sp.ent[at] = SmallEntry{uint32_t(p.node_off[v] + i) | (base + uint32_t(a * kv + i)) << 16,
                                        uint32_t(tnext) | uint32_t(m[v]) << 16 | 1u << 24};
                sp.ent_cpt[at] = p.cpt_flat[p.cpt_off[v] + int64_t(a) * kv + i];
                for (int jt = 0; jt < m[v]; ++jt) {
                    // the assignment without digit jt, same radix order: its rank among the assignments that share digit jt
                    int rest = 0;
                    for (int j = 0; j < m[v]; ++j)
                        if (j != jt) rest = rest * kp[j] + digit[j];
                    const uint32_t place = base + uint32_t(Sv) * uint32_t(1 + jt) + uint32_t((i * (rows[v] / kp[jt]) + rest) * kp[jt] + digit[jt]);
                    sp.term[tnext++] = uint32_t(p.msg_off[e0 + jt] + digit[jt]) | place << 16;
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
    for (int v = 0; v < n; ++v) {
        const TileDesc& td = p.tiles[p.node_tile[v]];
        const int nl = p.node_nl[v];
        for (int i = 0; i < p.k[v]; ++i) {
            // element i of pi(v) in the tile's striped node block (bn_tiles.hpp vidx(0, i, npt, nl))
            sp.nv_idx[p.node_off[v] + i] = int32_t(td.node_base + int64_t(i >> 1) * (int64_t(td.npt) * 2) + nl * 2 + (i & 1));
            sp.nv_slot[p.node_off[v] + i] = p.node_slot[v];
            if (m[v] == 0) sp.npi_init[p.node_off[v] + i] = p.cpt_flat[p.cpt_off[v] + i];  // a root starts from its CPT row (:58-64)
        }
    }
    sp.n = n; sp.N = int32_t(N); sp.M = int32_t(M); sp.S = int32_t(S); sp.T = int32_t(T); sp.TT = int32_t(TT);
    sp.CL = int32_t(E); sp.waves = waves; sp.mmax = mmax;
    // LDS: 4 M + 4 N + T doubles, TT words, CL halfwords, N marks, the residual words
    size_t bytes = size_t(4 * M + 4 * N + T) * 8 + ((size_t(std::max<int64_t>(TT, 1)) + 1) & ~size_t(1)) * 4;
    bytes += (size_t(std::max<int64_t>(E, 1)) * 2 + 7) & ~size_t(7);
    bytes += (size_t(N) + 7) & ~size_t(7);
    bytes += 2 * 16 * 8;  // the residual words: [iteration parity][wave]
    sp.lds_bytes = bytes;
    if (bytes > size_t(kSmallLdsBytes)) return no("state does not fit the LDS");
    sp.ok = true;
}

}  // namespace bnmi

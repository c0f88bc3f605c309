// bn_plan.cpp -- builds the lane-striped HBM layout from a flat model (host only, no HIP).
#include "bn_plan.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <unordered_map>

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace bnmi {

static inline int32_t round_even(int32_t x) { return (x + 1) & ~1; }

// Lanes per node.  k = 4 with 3, 4 or 5 parents: 4, 16 or 64 lanes cooperate on one node, each
// owning the 64 CPT entries of one assignment of the leading parents -- the same per-lane footprint
// as the register-resident path -- and the partial sums are combined with wave shuffles.
// Lanes per node of the lane-group variant (k = 4, 3..5 parents): 4^(m-2), 64 table entries per lane; `wide`
// (3 or 4 parents): 4^(m-1), 16 entries per lane -- four times the waves, a quarter of the serial work per wave.
static int pick_lanes(int kv, int m, bool all_k4, int forced, bool wide) {
    if (forced == 1) return 1;
    if (kv == 4 && all_k4 && m >= 3 && m <= 5) return 1 << (2 * (m - ((wide && m <= 4) ? 1 : 2)));
    return 1;
}

void default_owner(const bn_model_desc& d, int32_t nranks, std::vector<int32_t>& owner) {
    const int32_t n = d.n_nodes;
    owner.assign(std::max(n, 0), 0);
    if (n <= 0 || nranks <= 1) return;
    // weight = bytes a sweep moves for the node: its CPT plus a few k-vectors
    auto weight = [&](int32_t v) { return double(d.cpt_off[v + 1] - d.cpt_off[v]) + 8.0 * d.k[v]; };
    double total = 0;
    for (int32_t v = 0; v < n; ++v) total += weight(v);
    double cum = 0;
    for (int32_t v = 0; v < n; ++v) {
        int32_t r = int32_t(cum * nranks / total);
        owner[v] = std::min(std::max(r, 0), nranks - 1);
        cum += weight(v);
    }
}

// The lane-striped CPT image of the plan's tiles from a flat CPT array in the model's layout (bn_model_desc.cpt): into
// p.cpt_striped (p.cpt_doubles doubles).  Also what bn_reload_cpt re-runs when only the CPT values changed.
void stripe_cpt(Plan& p, const double* cpt) {
    const int32_t n = p.n;
    p.cpt_striped.assign(size_t(p.cpt_doubles), 0.0);
    // (every node writes its own lanes of its tile's run: chunks of nodes are independent)
    parallel_for(n, 4096, [&](int64_t v_begin, int64_t v_end) {
    for (int32_t v = int32_t(v_begin); v < int32_t(v_end); ++v) {
        if (p.node_class[v] < 0) continue;
        const ClassDesc& c = p.classes[p.node_class[v]];
        const TileDesc& td = p.tiles[p.node_tile[v]];
        const double* src = cpt + p.cpt_off[v];
        if (c.variant == kVariantFlat) {  // entry e of the reference's row-major table: lane e % G of the node's group, slot e / G
            double* dst = p.cpt_striped.data() + td.cpt_base;
            const int64_t S = int64_t(c.kv) * c.rows;
            for (int64_t e = 0; e < S; ++e) {
                const int64_t q = e / c.G, lane = int64_t(p.node_nl[v]) * c.G + e % c.G;
                dst[(q >> 1) * 128 + lane * 2 + (q & 1)] = src[e];
            }
            continue;
        }
        const int32_t cpl = c.rows / c.G;  // assignments per lane
        for (int g = 0; g < c.G; ++g) {
            const int lane = p.node_nl[v] * c.G + g;
            double* dst = p.cpt_striped.data() + td.cpt_base + lane * 2;
            for (int32_t i = 0; i < c.kv; ++i)
                for (int32_t cl = 0; cl < cpl; ++cl) {
                    const int32_t q = i * cpl + cl;
                    const int32_t cond = g * cpl + cl;
                    dst[int64_t(q >> 1) * 128 + (q & 1)] = src[int64_t(cond) * c.kv + i];
                }
        }
    }
    });
}

std::string build_plan(const bn_model_desc& d_in, const ShardSpec& shard, Plan& p) {
    // lanes_per_node: 0 automatic, 2 dense (automatic without the any-arity rule for nodes with many children),
    // 3 / 4 = 0 / 2 plus the wide lane-group split on small networks (bn_mi355x.h)
    bn_model_desc d = d_in;
    const bool plan_timing = std::getenv("BN_PLAN_TIMING") != nullptr;   // where build_plan's time goes, one line per section on stderr
    auto plan_t0 = std::chrono::steady_clock::now();
    auto plan_lap = [&](const char* what) {
        if (!plan_timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[bn_plan] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - plan_t0).count());
        plan_t0 = now;
    };
    const bool latency_rules = d_in.lanes_per_node == 0 || d_in.lanes_per_node == 3;
    const bool wide_requested = d_in.lanes_per_node == 3 || d_in.lanes_per_node == 4;
    if (d.lanes_per_node >= 2 && d.lanes_per_node <= 4) d.lanes_per_node = 0;
    const int32_t n = d.n_nodes;
    if (n < 0) return "n_nodes < 0";
    if (d_in.lanes_per_node < 0 || d_in.lanes_per_node > 4) return "lanes_per_node outside 0..4 (bn_mi355x.h)";
    if (n > 0 && (!d.k || !d.in_ptr || !d.cpt_off)) return "null model array";
    if (shard.nranks < 1 || shard.rank < 0 || shard.rank >= shard.nranks) return "bad rank / nranks";
    p = Plan();
    p.wide_requested = wide_requested;  // after the reset: dense_engine_for_batch (bn_engine_batch.cpp) reads it
    p.n = n;
    p.rank = shard.rank;
    p.nranks = shard.nranks;
    p.k.assign(d.k, d.k + n);
    if (d.in_ptr) p.in_ptr.assign(d.in_ptr, d.in_ptr + n + 1);
    else p.in_ptr.assign(1, 0);
    if (p.in_ptr[0] != 0) return "in_ptr[0] != 0";
    for (int32_t v = 0; v < n; ++v) {
        if (p.k[v] < 1 || p.k[v] > 255) return "selectable_num of node " + std::to_string(v) + " outside [1,255]";
        int32_t m = p.in_ptr[v + 1] - p.in_ptr[v];
        if (m < 0) return "in_ptr decreases at node " + std::to_string(v);
        if (m > BN_MAX_PARENTS)
            return "node " + std::to_string(v) + " has " + std::to_string(m) + " parents (max " +
                   std::to_string(BN_MAX_PARENTS) + ")";
    }
    p.E = p.in_ptr[n];
    if (p.E > 0 && !d.in_idx) return "null in_idx";
    p.in_idx.assign(d.in_idx, d.in_idx + p.E);
    p.node_off.assign(n + 1, 0);
    for (int32_t v = 0; v < n; ++v) p.node_off[v + 1] = p.node_off[v] + p.k[v];
    for (int32_t v = 0; v < n; ++v) {
        int64_t sz = p.k[v];
        for (int32_t e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e) {
            int32_t u = p.in_idx[e];
            if (u < 0 || u >= n) return "parent index out of range at node " + std::to_string(v);
            if (u == v) return "node " + std::to_string(v) + " is its own parent";
            if (e > p.in_ptr[v] && p.in_idx[e - 1] >= u)
                return "parents of node " + std::to_string(v) + " are not strictly ascending";
            sz *= p.k[u];
            if (sz > (int64_t(1) << 26)) return "CPT of node " + std::to_string(v) + " exceeds 2^26 entries";
        }
        // the reference would hit a missing row (UB, graph.hpp:120-124); here it is an error
        if (d.cpt_off[v + 1] - d.cpt_off[v] != sz)
            return "cpt_off of node " + std::to_string(v) + " does not match k * prod k[parents]";
    }
    if (n > 0 && d.cpt_off[0] != 0) return "cpt_off[0] != 0";
    if (n > 0 && d.cpt_off[n] > 0 && !d.cpt) return "null cpt";
    p.msg_off.assign(p.E + 1, 0);
    for (int64_t e = 0; e < p.E; ++e) p.msg_off[e + 1] = p.msg_off[e] + p.k[p.in_idx[e]];

    plan_lap("validate");
    // ---- ownership
    const int32_t me = p.rank;
    if (p.nranks > 1) {
        if (shard.owner) {
            p.owner.assign(shard.owner, shard.owner + n);
            for (int32_t v = 0; v < n; ++v)
                if (p.owner[v] < 0 || p.owner[v] >= p.nranks) return "owner[] entry out of range";
        } else {
            default_owner(d, p.nranks, p.owner);
        }
    }
    auto own = [&](int32_t v) { return p.nranks == 1 ? 0 : p.owner[v]; };
    std::vector<int32_t> edge_child(std::max<int64_t>(p.E, 1));
    for (int32_t v = 0; v < n; ++v)
        for (int32_t e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e) edge_child[e] = v;

    plan_lap("ownership");
    // ---- children (ascending) with the CSR edge id of each out-edge
    std::vector<int32_t> out_ptr(n + 1, 0), out_edge(std::max<int64_t>(p.E, 1));
    for (int64_t e = 0; e < p.E; ++e) out_ptr[p.in_idx[e] + 1]++;
    for (int32_t v = 0; v < n; ++v) out_ptr[v + 1] += out_ptr[v];
    {
        std::vector<int32_t> fill(n, 0);
        for (int32_t v = 0; v < n; ++v)
            for (int32_t e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e) {
                int32_t u = p.in_idx[e];
                out_edge[out_ptr[u] + fill[u]++] = e;
            }
    }

    // The lane-group split (automatic layout and lanes_per_node 3 / 4; not with 2 = dense: it re-associates the sums over
    // assignments, so its marginals differ from the 64-entries-per-lane split in the last bits -- both within rounding
    // of the reference, whose own >= 3-parent products are unordered -- and it quadruples the wavefronts a batch pays
    // for): a network whose lane-group tiles leave SIMDs idle is bound by the latency of ONE
    // tile (tile stamps: 4.2 of a 4-parent tile's 8.9 us are the contraction over its 64 entries per lane on a
    // SIMD that holds no other wave), so it takes the wide split while that still means at most one wave per
    // SIMD (measured, us per sweep, 64 -> 16 entries per lane: 200-node DAG 9.8 -> 8.6, 1000 nodes 10.1 -> 9.2,
    // 3000 nodes 10.3 -> 9.4; 10 k nodes = 2 699 tiles 11.9 -> 15.8: throughput-bound, keeps 64).  Decided on
    // the whole model: every rank alike.
    bool group_wide = false;
    int64_t est_tiles = 0;  // wavefronts of the dense layout, estimated before the classes exist
    {
        int64_t wide_tiles = 0, dense64 = 0;
        for (int32_t v = 0; v < n; ++v) {
            const int32_t kv = p.k[v], m = p.in_ptr[v + 1] - p.in_ptr[v];
            bool same = kv == 4 && m >= 3 && m <= 5;
            for (int32_t e = p.in_ptr[v]; e < p.in_ptr[v + 1] && same; ++e) same = p.k[p.in_idx[e]] == 4;
            if (same) wide_tiles += m == 3 ? 1 : 4;  // in quarters of a wave: 4 nodes per wave at m = 3, one at m = 4, 5
            dense64 += !same ? 1 : (m == 3 ? 4 : (m == 4 ? 16 : 64));  // in 64ths of a wave
        }
        wide_tiles /= 4;
        est_tiles = dense64 / 64;
        static const char* force = std::getenv("BN_GROUP_WIDE");  // A/B switch: 0 / 1
        // automatic layout (lanes_per_node 0): applied by itself, like the other latency rule -- with three or more
        // parents the reference's own products are unordered (:253), so neither split is "the" reference order
        group_wide = force ? force[0] == '1' : ((wide_requested || d_in.lanes_per_node == 0) && wide_tiles > 0 && wide_tiles <= 1024);
        p.group_wide = group_wide;
    }

    plan_lap("children");
    // ---- which nodes have a templated variant, which can take the any-arity variant
    auto shape_of = [&](int32_t v, bool& templated, bool& flat_ok) {
        const int32_t kv = p.k[v], m = p.in_ptr[v + 1] - p.in_ptr[v];
        bool same = kv >= 2 && kv <= 4 && m <= 4;
        int64_t S = kv, sum_kp = 0;
        for (int32_t e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e) {
            const int32_t ku = p.k[p.in_idx[e]];
            if (ku != kv) same = false;
            S *= ku;
            sum_kp += ku;
        }
        templated = (same && S <= 64) || (same && kv == 4 && m >= 3 && m <= 5 && d.lanes_per_node != 1);
        flat_ok = d.lanes_per_node != 1 && m <= kFlatMaxParents && sum_kp <= kWave && kv <= kWave && S < (int64_t(1) << 22);
        // A one-lane tile serves all children of a node on that node's lane: up to 4 fused with the child role,
        // beyond that in fully unrolled code that costs ~1 us per child (measured with the tile stamps: mostly
        // instruction fetch).  The any-arity tile gives every (child, state) a lane of its own, so a node with
        // more than 4 children goes there (its <= 64-entry table takes the ordered path: same bits) -- while the
        // network is small enough to be bound by the latency of one tile; it costs wavefronts.
        if (same && S <= 64 && flat_ok && latency_rules && est_tiles <= 1024 && out_ptr[v + 1] - out_ptr[v] > 4 &&
            int64_t(out_ptr[v + 1] - out_ptr[v]) * kv <= kWave) {
            templated = false;
            p.latency_rules_applied = true;
        }
    };
    // A network made mostly of any-arity tiles runs them all that way: the launch without
    // register-resident tiles has twice the occupancy, which is what those latency-bound tiles need.
    bool prefer_flat = false;
    if (d.lanes_per_node == 0) {
        int64_t n_templ = 0, n_flat_only = 0, n_convertible = 0;
        for (int32_t v = 0; v < n; ++v) {  // the whole model, not this rank's share: every rank decides alike
            bool t, f;
            shape_of(v, t, f);
            if (t) { ++n_templ; if (f) ++n_convertible; }
            else if (f) ++n_flat_only;
        }
        prefer_flat = n_flat_only >= 2 * n_templ && n_convertible == n_templ && n_flat_only > 0;
    }

    plan_lap("variants");
    // ---- shape classes over the owned nodes
    std::map<std::vector<int32_t>, int32_t> sig2cls;
    p.node_class.assign(n, -1);
    std::vector<int32_t> sig;
    for (int32_t v = 0; v < n; ++v) {
        if (own(v) != me) continue;
        ++p.n_owned;
        sig.clear();
        sig.push_back(p.k[v]);
        for (int32_t e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e) sig.push_back(p.k[p.in_idx[e]]);
        // flat tiles serve child (c, i) of the parent role on lane c * kv + i of the node's group, one pass per
        // G / kv children (each pass costs a dependent round trip: reference -> the child's record): a node with
        // many children gets a group wide enough for ONE pass, so the width wanted by the out-degree is part
        // of the class key (ignored by the other variants)
        int32_t g_children = 0;
        bool templated, flat_ok;
        shape_of(v, templated, flat_ok);
        if (prefer_flat && flat_ok) templated = false;
        if (!templated && flat_ok) {
            g_children = 8;
            while (g_children < kWave && int64_t(out_ptr[v + 1] - out_ptr[v]) > g_children / p.k[v]) g_children *= 2;
        }
        sig.push_back(g_children);
        auto it = sig2cls.find(sig);
        if (it == sig2cls.end()) {
            ClassDesc c;
            std::memset(&c, 0, sizeof c);
            c.kv = sig[0];
            c.m = int32_t(sig.size()) - 2;
            c.kvp = round_even(c.kv);
            int64_t rows = 1;
            bool uniform = (c.kv >= 2 && c.kv <= 4 && c.m <= 4);
            for (int j = 0; j < c.m; ++j) {
                c.kp[j] = sig[1 + j];
                c.kpp[j] = round_even(c.kp[j]);
                rows *= c.kp[j];
                if (c.kp[j] != c.kv) uniform = false;
            }
            for (int j = c.m - 1, s = 1; j >= 0; --j) { c.cstride[j] = s; s *= c.kp[j]; }
            c.rows = int32_t(rows);
            const bool all_k4 = uniform && c.kv == 4;
            if (int64_t(c.kv) * rows > 64) uniform = false;  // register-resident CPT: <= 64 entries
            c.G = pick_lanes(c.kv, c.m, all_k4, d.lanes_per_node, group_wide);
            c.variant = c.G > 1 ? kVariantGroup : (uniform ? kVariantUniform : kVariantGeneric);
            if (g_children > 0) { c.G = 1; c.variant = kVariantGeneric; }  // the any-arity variant was chosen above
            // every other shape: one wavefront per node (lanes_per_node == 1 keeps the
            // one-lane-per-node generic path, for A/B tests)
            int32_t sum_kp = 0;
            for (int j = 0; j < c.m; ++j) sum_kp += c.kp[j];
            if (g_children > 0) {
                c.variant = kVariantFlat;
                // the smallest group that holds the table at two entries per lane and every vector
                // in one register across the group; tables above 128 entries: the whole wave
                c.G = kWave;
                for (int g = 8; g < kWave; g *= 2)
                    if (int64_t(c.kv) * rows <= 2 * g && sum_kp <= g && c.kv <= g) { c.G = g; break; }
                c.G = std::max(c.G, g_children);
            }
            c.npt = kWave / c.G;
            c.flat_tab_off = -1;
            c.magic_kv = (65536 + c.kv - 1) / c.kv;
            c.magic_hv = ((1 << 20) + c.kvp / 2 - 1) / (c.kvp / 2);
            for (int j = 0; j < c.m && j < BN_MAX_PARENTS; ++j) c.lam_run[j] = int32_t(int64_t(c.kv) * rows / c.kp[j]);
            if (c.variant == kVariantFlat && int64_t(c.kv) * rows <= 2 * c.G && int64_t(c.kv) * rows <= 128) {
                // ordered path: digits and summation places of every entry, computed once here
                c.flat_tab_off = int32_t(p.flat_tab.size());
                const int32_t S = int32_t(c.kv * rows);
                for (int32_t e = 0; e < 2 * c.G; ++e) {
                    FlatEntry fe;
                    std::memset(&fe, 0, sizeof fe);
                    if (e < S) {
                        fe.valid = 1;
                        const int32_t cond = e / c.kv, ei = e % c.kv;
                        fe.ei = uint8_t(ei);
                        fe.pos_pi = uint8_t(ei * rows + cond);
                        int32_t dj[kFlatMaxParents] = {0};
                        for (int j = 0; j < c.m; ++j) { dj[j] = (cond / c.cstride[j]) % c.kp[j]; fe.dj[j] = uint8_t(dj[j]); }
                        for (int jt = 0; jt < c.m; ++jt) {
                            const int32_t kj = c.kp[jt], per_state = int32_t(rows / kj);
                            int32_t rest = 0;  // the assignment with digit jt removed, same radix order
                            for (int j = 0; j < c.m; ++j)
                                if (j != jt) rest += dj[j] * (j < jt ? c.cstride[j] / kj : c.cstride[j]);
                            fe.pos_lam[jt] = uint8_t(dj[jt] * (c.kv * per_state) + ei * per_state + rest);
                        }
                    }
                    p.flat_tab.push_back(fe);
                }
            }
            c.per_lane = int32_t((int64_t(c.kv) * rows + c.G - 1) / c.G);
            c.per_lane_pad = round_even(c.per_lane);
            int32_t off = 0;
            for (int j = 0; j < c.m; ++j) { c.rec_off[j] = off; off += 2 * c.kpp[j] * c.npt; }
            c.rec_doubles = off;
            it = sig2cls.emplace(sig, int32_t(p.classes.size())).first;
            p.classes.push_back(c);
        }
        p.node_class[v] = it->second;
        p.classes[it->second].n_nodes++;
        p.g_max = std::max(p.g_max, p.classes[it->second].G);
    }

    plan_lap("classes");
    // ---- tiles: per class, ascending node id, NPT nodes each; tile order by first node id
    struct ProtoTile { int32_t cls; std::vector<int32_t> nodes; bool boundary; };
    std::vector<ProtoTile> proto;
    {
        std::vector<int32_t> open(p.classes.size(), -1);
        for (int32_t v = 0; v < n; ++v) {
            int32_t c = p.node_class[v];
            if (c < 0) continue;
            if (open[c] < 0 || int32_t(proto[open[c]].nodes.size()) == p.classes[c].npt) {
                open[c] = int32_t(proto.size());
                proto.push_back({c, {}, false});
                proto.back().nodes.reserve(p.classes[c].npt);
            }
            proto[open[c]].nodes.push_back(v);
            for (int32_t e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e)
                if (own(p.in_idx[e]) != me) proto[open[c]].boundary = true;
        }
    }
    // Sharded plans: tiles none of whose nodes touches a cut edge ("interior": they never read the
    // exchange region) come first, the others last.  The launch of iteration s+1 over the interior
    // tiles needs nothing from the all-gather of iteration s and runs while it is in flight; only the
    // launch over the trailing tiles waits for it (bn_engine.cpp).  Stable: node order is kept inside
    // both groups, so neighbouring tiles stay neighbours.
    p.n_interior_tiles = int32_t(proto.size());
    if (shard.nranks > 1) {
        std::vector<uint8_t> touches(proto.size(), 0);
        for (size_t t = 0; t < proto.size(); ++t) {
            bool cut = proto[t].boundary;
            for (int32_t v : proto[t].nodes)
                for (int32_t q = out_ptr[v]; q < out_ptr[v + 1] && !cut; ++q) {
                    // the child of out-edge q: the node whose in-edge list holds CSR edge out_edge[q]
                    const int32_t e = out_edge[q];
                    const int32_t child = int32_t(std::upper_bound(p.in_ptr.begin(), p.in_ptr.end(), e) - p.in_ptr.begin()) - 1;
                    if (own(child) != me) cut = true;
                }
            touches[t] = cut ? 1 : 0;
        }
        std::vector<ProtoTile> ordered;
        ordered.reserve(proto.size());
        for (size_t t = 0; t < proto.size(); ++t)
            if (!touches[t]) ordered.push_back(std::move(proto[t]));
        p.n_interior_tiles = int32_t(ordered.size());
        for (size_t t = 0; t < proto.size(); ++t)
            if (touches[t]) ordered.push_back(std::move(proto[t]));
        proto.swap(ordered);
    }
    const int32_t nt = int32_t(proto.size());
    p.tiles.assign(nt, TileDesc());
    p.node_tile.assign(n, -1);
    p.node_nl.assign(n, -1);
    p.node_slot.assign(n, -1);
    int64_t cpt_cur = 0, rec_cur = 0, node_cur = 0, out_cur = 0, inref_cur = 0;
    int32_t slot_cur = 0;
    for (int32_t t = 0; t < nt; ++t) {
        const ClassDesc& c = p.classes[proto[t].cls];
        TileDesc& td = p.tiles[t];
        std::memset(&td, 0, sizeof td);
        td.cls = proto[t].cls;
        td.n_nodes = int32_t(proto[t].nodes.size());
        td.slot_base = slot_cur;
        td.cpt_base = cpt_cur;
        td.rec_base = rec_cur;
        td.node_base = node_cur;
        td.out_base = out_cur;
        td.in_ref_base = proto[t].boundary ? inref_cur : -1;
        td.kv = uint8_t(c.kv); td.m = uint8_t(c.m); td.variant = uint8_t(c.variant); td.npt = uint8_t(c.npt);
        int32_t cmax = 0;
        for (int32_t nl = 0; nl < td.n_nodes; ++nl) {
            int32_t v = proto[t].nodes[nl];
            p.node_tile[v] = t;
            p.node_nl[v] = nl;
            p.node_slot[v] = slot_cur + nl;
            cmax = std::max(cmax, out_ptr[v + 1] - out_ptr[v]);
        }
        td.cmax = cmax;
        cpt_cur += int64_t(c.per_lane_pad) * kWave;
        rec_cur += c.rec_doubles;
        node_cur += int64_t(2) * c.kvp * c.npt;
        out_cur += int64_t(cmax) * c.npt;
        if (proto[t].boundary) inref_cur += int64_t(c.m) * c.npt;
        slot_cur += c.npt;
    }
    p.light = nt > 0;
    p.variants = 0;
    for (const ClassDesc& c : p.classes) {
        if (c.variant == kVariantUniform || c.variant == kVariantGroup) p.light = false;
        p.variants |= 1 << c.variant;
    }
    p.n_slots = slot_cur;
    p.rec_doubles = rec_cur;
    p.node_doubles = node_cur;
    p.slot_node.assign(slot_cur, -1);
    p.slot_boff.assign(slot_cur, -1);
    for (int32_t v = 0; v < n; ++v)
        if (p.node_slot[v] >= 0) {
            p.slot_node[p.node_slot[v]] = v;
            p.slot_boff[p.node_slot[v]] = p.node_off[v];
        }

    plan_lap("tiles");
    // ---- exchange region: every rank derives the same segment layout from the global graph.
    // Halves are enumerated in CSR edge order; the pi-half of a cut edge belongs to the parent's
    // owner, the lambda-half to the child's owner.
    p.seg_used_d2.assign(p.nranks, 0);
    std::unordered_map<int64_t, std::pair<int64_t, int64_t>> cut_off;  // e -> (pi off, lambda off), incident only
    if (p.nranks > 1) {
        for (int64_t e = 0; e < p.E; ++e) {
            const int32_t u = p.in_idx[e], v = edge_child[e];
            const int32_t a = own(u), b = own(v);
            if (a == b) continue;
            const int64_t h = round_even(p.k[u]) / 2;
            const int64_t po = p.seg_used_d2[a];
            p.seg_used_d2[a] += h;
            const int64_t lo = p.seg_used_d2[b];
            p.seg_used_d2[b] += h;
            if (a == me || b == me) {
                cut_off.emplace(e, std::make_pair(po, lo));
                ++p.n_cut_edges;
            }
        }
    }
    int64_t seg_data = 0;
    for (int64_t s : p.seg_used_d2) seg_data = std::max(seg_data, s);
    seg_data = (seg_data + 7) & ~int64_t(7);  // 128-byte granularity
    p.seg_data_d2 = seg_data;
    p.seg_d2 = seg_data + kResSlotsD2;
    p.g_base = rec_cur / 2;
    p.rec_total_doubles = rec_cur + 2 * p.seg_d2 * p.nranks;
    if (cpt_cur / 2 > INT32_MAX || p.rec_total_doubles / 2 > INT32_MAX) return "model too large for 32-bit record indices";

    plan_lap("exchange");
    // ---- CPT image: i-major per node, assignments split over the G lanes of the node
    p.cpt_doubles = cpt_cur;   // (stripe_cpt sizes and fills p.cpt_striped)
    if (n > 0) p.cpt_off.assign(d.cpt_off, d.cpt_off + n + 1);
    else p.cpt_off.assign(1, 0);
    p.cpt_flat.assign(d.cpt, d.cpt + (n ? d.cpt_off[n] : 0));
    plan_lap("cpt flat copy");
    stripe_cpt(p, d.cpt);

    plan_lap("cpt image");
    // ---- where each edge's messages live on this rank
    auto seg_index = [&](int32_t r, int64_t off) { return int32_t(p.g_base + int64_t(r) * p.seg_d2 + off); };
    p.edge_ref.assign(std::max<int64_t>(p.E, 1), MsgRef{-1, 0});
    for (int64_t e = 0; e < p.E; ++e) {
        const int32_t u = p.in_idx[e], v = edge_child[e];
        const int32_t a = own(u), b = own(v);
        if (a != me && b != me) continue;
        if (a == b) {  // both endpoints here: the record sits in the child's tile
            const ClassDesc& c = p.classes[p.node_class[v]];
            const TileDesc& td = p.tiles[p.node_tile[v]];
            const int32_t j = int32_t(e - p.in_ptr[v]);
            const int32_t pi = int32_t((td.rec_base + c.rec_off[j]) / 2 + p.node_nl[v]);
            p.edge_ref[e] = MsgRef{pi, pi + (c.kpp[j] / 2) * c.npt};
        } else {  // cut edge: pi-half in the parent owner's segment, lambda-half in the child owner's
            const auto& off = cut_off.at(e);
            p.edge_ref[e] = MsgRef{seg_index(a, off.first), ~seg_index(b, off.second)};
        }
    }
    p.out_refs.assign(std::max<int64_t>(out_cur, 1), MsgRef{-1, 0});
    p.in_refs.assign(std::max<int64_t>(inref_cur, 1), MsgRef{-1, 0});
    for (int32_t v = 0; v < n; ++v) {
        if (p.node_class[v] < 0) continue;
        const ClassDesc& c = p.classes[p.node_class[v]];
        const TileDesc& td = p.tiles[p.node_tile[v]];
        for (int32_t q = out_ptr[v]; q < out_ptr[v + 1]; ++q)
            p.out_refs[td.out_base + int64_t(q - out_ptr[v]) * c.npt + p.node_nl[v]] = p.edge_ref[out_edge[q]];
        if (td.in_ref_base >= 0)
            for (int32_t j = 0; j < c.m; ++j)
                p.in_refs[td.in_ref_base + int64_t(j) * c.npt + p.node_nl[v]] = p.edge_ref[p.in_ptr[v] + j];
    }

    plan_lap("edge refs");
    // ---- neighbour tiles (dataflow form of the resident kernel): the tiles of the parents and children of a tile's
    // nodes on this rank, ascending, without the tile itself; across a cut edge the peer's node stands in for its
    // tile until the peer's export blob names it (bn_engine_shard.cpp: bn_peer_import)
    {
        std::vector<std::vector<int32_t>> nb(nt);
        std::vector<uint8_t> is_boundary(n, 0);
        p.cut_links.clear();
        for (int64_t e = 0; e < p.E; ++e) {
            const int32_t u = p.in_idx[e], v = edge_child[e];
            const int32_t tu = p.node_tile[u], tv = p.node_tile[v];
            if (tu >= 0 && tv >= 0) {
                if (tu != tv) { nb[tu].push_back(tv); nb[tv].push_back(tu); }
            } else if (tu >= 0) {  // the child lives on another rank
                is_boundary[u] = 1;
                p.cut_links.push_back({tu, own(v), v});
            } else if (tv >= 0) {  // the parent does
                is_boundary[v] = 1;
                p.cut_links.push_back({tv, own(u), u});
            }
        }
        p.boundary_node.clear();
        p.boundary_tile.clear();
        for (int32_t v = 0; v < n; ++v)
            if (is_boundary[v]) { p.boundary_node.push_back(v); p.boundary_tile.push_back(p.node_tile[v]); }
        p.nbl_ptr.assign(nt + 1, 0);
        p.nbl_idx.clear();
        for (int32_t t = 0; t < nt; ++t) {
            std::sort(nb[t].begin(), nb[t].end());
            nb[t].erase(std::unique(nb[t].begin(), nb[t].end()), nb[t].end());
            p.nbl_idx.insert(p.nbl_idx.end(), nb[t].begin(), nb[t].end());
            p.nbl_ptr[t + 1] = int32_t(p.nbl_idx.size());
        }
        std::vector<std::vector<int32_t>> none(nt);
        std::string err = build_neighbour_table(p, none);
        (void)err;  // too many neighbours: nbr stays empty, the plan is not eligible for the dataflow form
    }

    plan_lap("neighbour tiles");
    // ---- metrics, this rank's share: CPT of owned nodes once; every message it
    // produces and every owned node vector read once and written once
    int64_t vec = 0, cpt_owned = 0, msgs = 0;
    for (int64_t e = 0; e < p.E; ++e) {
        const int32_t u = p.in_idx[e], v = edge_child[e];
        if (own(u) == me) { vec += p.k[u]; ++msgs; }
        if (own(v) == me) { vec += p.k[u]; ++msgs; }
    }
    for (int32_t v = 0; v < n; ++v)
        if (own(v) == me) { vec += 2 * int64_t(p.k[v]); cpt_owned += d.cpt_off[v + 1] - d.cpt_off[v]; }
    p.algorithmic_bytes = 8 * cpt_owned + 16 * vec;
    p.messages_per_sweep = msgs;
    // what the sweep kernel requests: CPT image, each record read by both endpoints and each half
    // written once, node vectors read + written, message references, tile descriptors, flags
    p.layout_bytes = 8 * cpt_cur + 8 * (2 * rec_cur + rec_cur) + 8 * 2 * node_cur + 8 * out_cur + 8 * inref_cur +
                     int64_t(sizeof(TileDesc)) * nt + slot_cur;
    return "";
}

// Plan::nbr from the local neighbour lists plus, per tile, the slots of its neighbour tiles on other ranks.
std::string build_neighbour_table(Plan& p, const std::vector<std::vector<int32_t>>& remote_slots) {
    const int32_t nt = int32_t(p.tiles.size());
    const int32_t base = p.rank * kFlowSlotsPerRank;  // local tile t has slot rank * kFlowSlotsPerRank + t
    p.nbr.clear();
    p.nbr_max = 0;
    for (int32_t t = 0; t < nt; ++t)
        p.nbr_max = std::max(p.nbr_max, p.nbl_ptr[t + 1] - p.nbl_ptr[t] + int32_t(remote_slots[t].size()));
    p.nbr_chunks = std::max(1, (p.nbr_max + kWave - 1) / kWave);
    if (p.nbr_chunks > kMaxNbrChunks) return "a tile has more than " + std::to_string(kMaxNbrChunks * kWave) + " neighbour tiles";
    if (nt > kFlowSlotsPerRank) return "more tiles than granule slots per rank";
    const size_t per_tile = size_t(p.nbr_chunks) * kWave;
    p.nbr.assign(size_t(std::max(nt, 1)) * per_tile, -1);
    for (int32_t t = 0; t < nt; ++t) {
        int32_t* row = p.nbr.data() + size_t(t) * per_tile;
        int32_t k = 0;
        for (int32_t q = p.nbl_ptr[t]; q < p.nbl_ptr[t + 1]; ++q) row[k++] = base + p.nbl_idx[q];
        for (int32_t slot : remote_slots[t]) row[k++] = slot;
    }
    return "";
}

void unstripe_messages(const Plan& p, const std::vector<double>& rec, double* pi_msg, double* lambda_msg) {
    for (int64_t e = 0; e < p.E; ++e) {
        const MsgRef r = p.edge_ref[e];
        if (r.pi < 0) continue;
        const int32_t ku = p.k[p.in_idx[e]];
        const int32_t h = ((ku + 1) & ~1) / 2;  // chunks per message
        const bool cut = r.lam < 0;
        const int64_t lam = cut ? ~r.lam : r.lam;
        const int64_t stride = cut ? 1 : (lam - r.pi) / h;
        for (int32_t i = 0; i < ku; ++i) {
            pi_msg[p.msg_off[e] + i] = rec[(int64_t(r.pi) + int64_t(i >> 1) * stride) * 2 + (i & 1)];
            lambda_msg[p.msg_off[e] + i] = rec[(lam + int64_t(i >> 1) * stride) * 2 + (i & 1)];
        }
    }
}

}  // namespace bnmi

// bn_dag_plan.cpp -- host plan of the register-resident path for networks of arity <= 4 with up to 5 parents per node (bn_dag.hpp).
// Arities below 4 are PADDED to 4: a node's table is laid out as if it and its parents had four states, with zeros wherever a state
// does not exist.  A zero entry adds +0.0 to a sum and makes a product 0: the real entries of every vector keep the reference's bits,
// the padding entries stay 0 -- provided the run STARTS with zeros there, which is why such a network's initial state is written
// to memory before a run (DagPlan::uniform4 == false, dag_init_kernel) instead of being synthesised as all-ones in registers.
#include "bn_dag.hpp"

#include <algorithm>
#include <functional>
#include <numeric>
#include <queue>

namespace bnmi {

namespace {
inline int lanes_of(int m) { return m <= 2 ? 1 : 1 << (2 * (m - 2)); }          // lanes that share a node
inline int entries_of(int m) { return m == 0 ? 4 : (m == 1 ? 16 : 64); }        // CPT entries per lane
// what a tile costs its SIMD per iteration, relative to a 4-parent child tile (instruction counts of bn_dag.hip)
inline double cost_of(const DagTile& t) {
    switch (t.kind) {
        case 0: return 0.10;
        case 1: return 0.30;
        case 2: return 0.85;   // 64 entries in the reference's order
        case 3: return 0.90;
        case 4: return 1.00;
        case 5: return 1.10;
        default: return 0.25 + 0.03 * t.dmax;   // parent items
    }
}
}  // namespace

void build_dag_plan(const Plan& p, int32_t cap_blocks, DagPlan& dp, bool light) {
    dp = DagPlan();
    dp.light = light;
    const int32_t n = p.n;
    if (p.nranks != 1) { dp.why = "sharded engine"; return; }
    if (n < 1) { dp.why = "empty network"; return; }
    if (p.E + int64_t(n) >= (int64_t(1) << 23)) { dp.why = "state beyond 32-bit byte offsets"; return; }
    for (int32_t v = 0; v < n; ++v) {
        if (p.k[v] > 4) { dp.why = "a node's arity is above 4"; return; }
        if (p.k[v] != 4) dp.uniform4 = false;
        if (p.in_ptr[v + 1] - p.in_ptr[v] > kDagMaxParents) { dp.why = "a node has more than 5 parents"; return; }
    }
    cap_blocks = std::min<int32_t>(cap_blocks & ~7, kDagMaxBlocks);
    if (cap_blocks < 8) { dp.why = "device too small"; return; }
    {
        // The padded register image: 4^(m+1) entries per node whatever its arities.  Beyond kDagMaxImageBytes the path is refused: in
        // stream form the image is re-read every sweep (a padded binary network with 5-parent nodes is 64x its model), the tile
        // kernels read the unpadded tables at the HBM rate, and DagTile::cpt_base is a 32-bit double2 index.
        int64_t img_entries = 0;
        for (int32_t v = 0; v < n; ++v) {
            const int m = p.in_ptr[v + 1] - p.in_ptr[v];
            img_entries += int64_t(4) << (2 * m);   // (tiles are padded to 64 lanes: at most one tile per parent count more)
        }
        if (img_entries * 8 > kDagMaxImageBytes) { dp.why = "padded CPT image beyond 256 MB"; return; }
    }
    dp.n = n;
    dp.E = int32_t(p.E);

    // children (ascending) with the CSR edge id of each out-edge
    std::vector<int32_t> out_ptr(n + 1, 0);
    for (int64_t e = 0; e < p.E; ++e) out_ptr[p.in_idx[e] + 1]++;
    for (int32_t v = 0; v < n; ++v) out_ptr[v + 1] += out_ptr[v];
    for (int32_t v = 0; v < n; ++v)
        if (out_ptr[v + 1] - out_ptr[v] > kDagMaxChildren) {
            // DagParentLane packs child count | rank << 16, and the items of a node with more than 63 children each walk ALL its
            // children (deg^2 record loads per sweep): hubs beyond this stay on the tile / item kernels
            dp = DagPlan();
            dp.why = "a node has more than " + std::to_string(kDagMaxChildren) + " children";
            return;
        }
    dp.oedge.assign(std::max<int64_t>(p.E, 1), 0);
    {
        std::vector<int32_t> fill(n, 0);
        for (int32_t v = 0; v < n; ++v)   // children visited in ascending order: every list comes out ascending
            for (int32_t e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e) {
                const int32_t u = p.in_idx[e];
                dp.oedge[out_ptr[u] + fill[u]++] = e;
            }
    }
    // initial pi(v): a root's CPT row (:58-64, not normalised), else ones -- over the node's OWN states, 0 in the padding
    dp.npi_init.assign(size_t(n) * 4, 0.0);
    for (int32_t v = 0; v < n; ++v)
        for (int i = 0; i < p.k[v]; ++i)
            dp.npi_init[size_t(v) * 4 + i] = p.in_ptr[v + 1] == p.in_ptr[v] ? p.cpt_flat[p.cpt_off[v] + i] : 1.0;

    // ---- tiles in natural order: child tiles by parent count, then parent items by child count
    std::vector<DagTile> tiles;
    std::vector<DagChildLane> cnode;
    std::vector<DagParentLane> pitem;
    BigVec img;
    size_t img_doubles = 0;
    auto new_tile = [&](int32_t kind) -> DagTile& {
        DagTile t{};
        t.kind = kind;
        t.lane_base = int32_t(tiles.size()) * kWave;
        tiles.push_back(t);
        cnode.resize(tiles.size() * kWave, DagChildLane{-1, 0});
        pitem.resize(tiles.size() * kWave, DagParentLane{-1, -1, 0, 0});
        return tiles.back();
    };
    for (int m = 0; m <= kDagMaxParents; ++m) {
        const int G = lanes_of(m), npt = kWave / G, epl = entries_of(m);
        std::vector<int32_t> nodes;
        for (int32_t v = 0; v < n; ++v)
            if (p.in_ptr[v + 1] - p.in_ptr[v] == m) nodes.push_back(v);
        for (size_t at = 0; at < nodes.size(); at += size_t(npt)) {
            DagTile& t = new_tile(m);
            t.n_active = int32_t(std::min<size_t>(npt, nodes.size() - at));
            t.cpt_base = int32_t(img_doubles / 2);
            img_doubles += size_t(epl) * kWave;
            for (int nl = 0; nl < t.n_active; ++nl)
                for (int g = 0; g < G; ++g) cnode[size_t(t.lane_base) + nl * G + g] = DagChildLane{nodes[at + nl], p.in_ptr[nodes[at + nl]]};
        }
    }
    if (!light) {
        // The image (light plans: the policy's features and the tile tables only -- the image is filled when the path is first used).
        // Every child tile fills its own run: tiles are independent (parallel_for).
        img.assign(img_doubles, 0.0);
        const int64_t n_child = int64_t(tiles.size());
        parallel_for(n_child, 64, [&](int64_t t_begin, int64_t t_end) {
            for (int64_t ti = t_begin; ti < t_end; ++ti) {
                const DagTile& t = tiles[size_t(ti)];
                const int m = t.kind, G = lanes_of(m);
                double* im = img.data() + size_t(t.cpt_base) * 2;
                for (int nl = 0; nl < t.n_active; ++nl) {
                    const int32_t v = cnode[size_t(t.lane_base) + size_t(nl) * G].node;
                    const double* cpt = p.cpt_flat.data() + p.cpt_off[v];
                    for (int g = 0; g < G; ++g) {
                        const int lane = nl * G + g;
                        // the lane's leading-parent digits (first parent most significant) are the digits of g; its entries run
                        // over the trailing parents (the last one fastest) and the own state: entry cl * 4 + i
                        const int trailing = m < 2 ? m : 2, ncl = 1 << (2 * trailing);
                        const int kv = p.k[v];
                        for (int cl = 0; cl < ncl; ++cl)
                            for (int i = 0; i < 4; ++i) {
                                // the assignment in base 4, first parent most significant; the entry it names in the node's REAL table
                                // (mixed radix of the parents' arities), or none: a parent state or an own state that does not exist
                                const int64_t row4 = (int64_t(g) << (2 * trailing)) | cl;
                                int64_t row = 0;
                                bool real = i < kv;
                                for (int j = 0; j < m; ++j) {
                                    const int digit = int((row4 >> (2 * (m - 1 - j))) & 3);
                                    const int kj = p.k[p.in_idx[p.in_ptr[v] + j]];
                                    real = real && digit < kj;
                                    row = row * kj + digit;
                                }
                                const int q = cl * 4 + i;
                                im[size_t(q >> 1) * (2 * kWave) + size_t(lane) * 2 + (q & 1)] = real ? cpt[row * kv + i] : 0.0;
                            }
                    }
                }
            }
        });
    }
    dp.n_child_tiles = int32_t(tiles.size());
    {
        double padded = 0.0;
        for (int32_t v = 0; v < n; ++v) padded += double(int64_t(4) << (2 * (p.in_ptr[v + 1] - p.in_ptr[v])));
        dp.fill = padded > 0.0 ? double(p.cpt_off[n]) / padded : 1.0;
    }
    for (const DagTile& t : tiles) dp.has_groups = dp.has_groups || t.kind >= 3;
    {
        // Parent items, nodes in order of their child count.  The c + 1 items of a node (lambda(v), then the pi-message to each
        // child in ascending order) sit in ADJACENT lanes of ONE wave: every lane loads one record -- pi(v), or the lambda-message
        // of its own child -- and the node's lanes exchange them through the wave's LDS scratch (kDagParent).  A node with more
        // than 63 children does not fit a wave: its items load every record themselves (kDagParentWide).
        std::vector<int32_t> order(n);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            return out_ptr[a + 1] - out_ptr[a] < out_ptr[b + 1] - out_ptr[b];
        });
        int fill = kWave;
        int32_t cur = -1, cur_kind = -1;
        auto push = [&](int32_t kind, const DagParentLane& it) {
            if (fill == kWave || kind != cur_kind) { new_tile(kind); cur = int32_t(tiles.size()) - 1; cur_kind = kind; fill = 0; }
            pitem[size_t(tiles[cur].lane_base) + fill++] = it;
            tiles[cur].n_active = fill;
            tiles[cur].dmax = std::max(tiles[cur].dmax, it.deg_tpos & 0xffff);
        };
        for (int32_t u : order) {
            const int32_t deg = out_ptr[u + 1] - out_ptr[u];
            const int32_t kind = deg + 1 <= kWave ? kDagParent : kDagParentWide;
            if (kind == kDagParent && fill + deg + 1 > kWave) fill = kWave;   // the node would straddle two waves: start a new one
            push(kind, DagParentLane{u, -1, out_ptr[u], int32_t(uint32_t(deg) | 0xffff0000u)});   // lambda(v): no child is skipped
            for (int32_t r = 0; r < deg; ++r) push(kind, DagParentLane{u, dp.oedge[out_ptr[u] + r], out_ptr[u], deg | (r << 16)});
        }
    }
    dp.n_parent_tiles = int32_t(tiles.size()) - dp.n_child_tiles;

    // ---- wave slots.  One block per CU, kDagWaves waves: waves w and w + 4 share a SIMD.  Blocks: enough for every SIMD's load
    // to stay near ONE heavy tile's (the iteration's critical path is the slowest SIMD), at most cap_blocks.
    const int64_t total = int64_t(tiles.size());
    double cost_sum = 0.0;
    for (const DagTile& t : tiles) cost_sum += cost_of(t);
    int64_t want = std::max<int64_t>((total + kDagWaves - 1) / kDagWaves, int64_t(cost_sum / 0.9 / 4.0) + 1);
    want = (std::max<int64_t>(want, 1) + 7) & ~int64_t(7);
    if (total <= kDagWaves) want = 1;   // one block: no grid barrier at all
    const int32_t nb = int32_t(std::min<int64_t>(want, cap_blocks));
    const int32_t slots = nb * kDagWaves;
    dp.blocks = nb;
    dp.stream = total > slots;
    // Longest-processing-time-first: tiles in order of decreasing cost, each to the least loaded SIMD that still has a free wave
    // (two per SIMD; stream form: to the least loaded wave).  SIMDs are numbered SIMD-major (SIMD 0 of every block, then SIMD 1
    // of every block, ...), so that the heavy tiles -- placed first, onto empty SIMDs -- spread over all blocks' CUs.
    std::vector<int32_t> by_cost(tiles.size());
    std::iota(by_cost.begin(), by_cost.end(), 0);
    std::stable_sort(by_cost.begin(), by_cost.end(), [&](int32_t a, int32_t b) { return cost_of(tiles[a]) > cost_of(tiles[b]); });
    std::vector<std::vector<int32_t>> of_slot(slots);
    {
        typedef std::pair<double, int32_t> Bin;   // (load, bin)
        std::priority_queue<Bin, std::vector<Bin>, std::greater<Bin>> pq;
        const int32_t nbins = dp.stream ? slots : nb * 4;
        std::vector<int32_t> used(nbins, 0);
        for (int32_t q = 0; q < nbins; ++q) pq.push(Bin(0.0, q));
        for (int32_t t : by_cost) {
            const Bin top = pq.top();
            pq.pop();
            const int32_t q = top.second;
            int32_t slot;
            if (dp.stream) {   // bin = wave: wave w of every block, then wave w + 1, ...
                slot = (q % nb) * kDagWaves + q / nb;
                pq.push(Bin(top.first + cost_of(tiles[t]), q));
            } else {           // bin = SIMD: its first wave, then the one four further
                slot = (q % nb) * kDagWaves + q / nb + 4 * used[q];
                if (++used[q] < kDagWaves / 4) pq.push(Bin(top.first + cost_of(tiles[t]), q));
            }
            of_slot[slot].push_back(t);
        }
    }
    dp.slot_ptr.assign(size_t(slots) + 1, 0);
    dp.tiles.reserve(tiles.size());
    dp.cnode.resize(tiles.size() * kWave);
    dp.pitem.resize(tiles.size() * kWave);
    for (int32_t s = 0; s < slots; ++s) {
        for (int32_t t : of_slot[s]) {
            DagTile nt = tiles[t];
            nt.lane_base = int32_t(dp.tiles.size()) * kWave;
            std::copy(cnode.begin() + size_t(tiles[t].lane_base), cnode.begin() + size_t(tiles[t].lane_base) + kWave, dp.cnode.begin() + nt.lane_base);
            std::copy(pitem.begin() + size_t(tiles[t].lane_base), pitem.begin() + size_t(tiles[t].lane_base) + kWave, dp.pitem.begin() + nt.lane_base);
            dp.tiles.push_back(nt);
        }
        dp.slot_ptr[s + 1] = int32_t(dp.tiles.size());
    }
    dp.cpt_img.swap(img);
    dp.ok = true;
}

// The lane tables as the kernels read them: CSR ids replaced by state records, numbered tile-major (bn_dag.hpp).
void build_dag_device_tables(const DagPlan& dp, DagDeviceTables& dt) {
    dt = DagDeviceTables();
    dt.eperm.assign(size_t(dp.E), -1);
    dt.nperm.assign(size_t(dp.n), -1);
    dt.tiles = dp.tiles;
    int32_t next_rec = 0, next_slot = 0;
    for (DagTile& t : dt.tiles) {
        if (t.kind >= kDagParent) continue;
        t.rec_base = next_rec;
        t.slot_base = next_slot;
        const int M = t.kind;
        const int G = M <= 2 ? 1 : 1 << (2 * (M - 2));   // lanes per node
        const int act = t.n_active;                      // nodes of the tile: lane groups 0 .. act - 1
        for (int i = 0; i < act; ++i) {
            const DagChildLane& cl = dp.cnode[size_t(t.lane_base) + size_t(i) * G];
            dt.nperm[size_t(cl.node)] = next_slot + i;
            for (int j = 0; j < M; ++j) dt.eperm[size_t(cl.ebase) + j] = next_rec + j * act + i;
        }
        next_rec += M * act;
        next_slot += act;
    }
    // (every node sits in exactly one child tile, every edge is an in-edge of one node: both numberings are complete)
    dt.pitem.resize(dp.pitem.size());
    for (size_t l = 0; l < dp.pitem.size(); ++l) {
        const DagParentLane& it = dp.pitem[l];
        DagParentLaneDev d{it.node, it.tedge, it.obeg, it.deg_tpos, 0, {0, 0, 0}};
        if (it.node >= 0) {
            d.snode = dt.nperm[size_t(it.node)];
            if (it.tedge >= 0) d.tedge = dt.eperm[size_t(it.tedge)];
        }
        dt.pitem[l] = d;
    }
    dt.oedge.resize(dp.oedge.size());
    for (size_t x = 0; x < dp.oedge.size(); ++x) dt.oedge[x] = dp.E > 0 ? dt.eperm[size_t(dp.oedge[x])] : 0;   // (a network without edges keeps one unused entry)
}

// The tiles a tile exchanges messages with (dataflow form of bp_dag_kernel, bn_dag.hip): a child tile reads the pi-messages its
// nodes' parents' items produce and the lambda(v) its nodes' own items produce; a parent tile reads pi(u) of its nodes and the
// lambda-messages of their children -- the relation is symmetric (who reads from a tile is read by it), so ONE list per tile
// covers both what it waits for (its inputs) and whom it must not overtake (the readers of what it overwrites).
// nbr [tiles][kWave], -1 = none; ok = false when some tile has more than kWave neighbours (a lane polls one neighbour), when the
// plan is in stream form (several tiles per wave) or runs in one block (no barrier to replace).
void build_dag_flow_tables(const DagPlan& dp, const Plan& p, DagFlowTables& ft) {
    ft = DagFlowTables();
    if (!dp.ok || dp.light || dp.stream || dp.blocks < 2) return;
    const size_t nt = dp.tiles.size();
    const int32_t n = dp.n;
    std::vector<int32_t> ctile(size_t(n), -1);                 // the child tile of a node
    std::vector<std::vector<int32_t>> ptiles{size_t(n)};       // the parent tile(s) that hold a node's items (several: > 63 children)
    for (size_t t = 0; t < nt; ++t) {
        const DagTile& td = dp.tiles[t];
        for (int l = 0; l < kWave; ++l) {
            if (td.kind < kDagParent) {
                const int32_t v = dp.cnode[size_t(td.lane_base) + l].node;
                if (v >= 0) ctile[size_t(v)] = int32_t(t);
            } else {
                const int32_t u = dp.pitem[size_t(td.lane_base) + l].node;
                if (u >= 0 && (ptiles[size_t(u)].empty() || ptiles[size_t(u)].back() != int32_t(t))) ptiles[size_t(u)].push_back(int32_t(t));
            }
        }
    }
    std::vector<std::vector<int32_t>> nbr(nt);
    auto link = [&](int32_t a, int32_t b) { nbr[size_t(a)].push_back(b); nbr[size_t(b)].push_back(a); };
    for (int32_t v = 0; v < n; ++v) {
        const int32_t c = ctile[size_t(v)];
        if (c < 0) return;   // (every node sits in a child tile: a plan that says otherwise is not one to run)
        for (int32_t t : ptiles[size_t(v)]) link(c, t);
        for (int32_t e = p.in_ptr[v]; e < p.in_ptr[v + 1]; ++e)
            for (int32_t t : ptiles[size_t(p.in_idx[e])]) link(c, t);
    }
    ft.nbr.assign(nt * kWave, -1);
    for (size_t t = 0; t < nt; ++t) {
        std::vector<int32_t>& l = nbr[t];
        std::sort(l.begin(), l.end());
        l.erase(std::unique(l.begin(), l.end()), l.end());
        ft.max_nbr = std::max<int32_t>(ft.max_nbr, int32_t(l.size()));
        if (l.size() > size_t(kWave)) { ft.nbr.clear(); return; }
        std::copy(l.begin(), l.end(), ft.nbr.begin() + t * kWave);
    }
    ft.ok = true;
}

}  // namespace bnmi

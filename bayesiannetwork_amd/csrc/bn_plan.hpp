// bn_plan.hpp -- host-side layout plan: how a flat model (bn_model_desc) is laid out in HBM.
//
// The reference keeps a CPT as unordered_map<condition_t, vector<double>> (graph.hpp:57-154)
// and BP state in eight hash maps (belief_propagation.hpp:320-333).  Here everything is
// lane-striped for 64-wide wavefronts so that every wave-level load is one contiguous run:
//
//   tile      = one wavefront's worth of nodes of ONE shape class (same k, same parent arities),
//               NPT = 64 / G nodes, G = lanes cooperating on one node
//               (G = 1: register-resident / generic; 4..64: lane groups for k = 4; 64 with the flat
//               row-major image: any arities, see tile_flat)
//   CPT       = per tile, i-major (own state slowest, parent assignment fastest) so that one
//               streaming pass visits entries in the reference's accumulation order for both
//               calculate_pi (:174-200) and calculate_lambda_k (:240-266); lane (node nl, part g)
//               owns assignments [g*C/G, (g+1)*C/G); its q-th entry sits at
//               cpt_base + (q/2)*128 + lane*2 + (q&1)      -> 16 B per lane, 1 KiB per wave load
//   records   = per in-edge (child tile, parent slot j, node nl): the pi-message then the
//               lambda-message of that edge, each Kp = roundup(k[parent], 2) doubles, striped as
//               rec + chunk*(NPT*2) + nl*2 ; the child reads/writes them fully coalesced, the
//               parent reaches them through a per-out-edge MsgRef
//   node vecs = pi(v) then lambda(v), striped the same way per tile
//
// All state that changes per sweep (records, node vectors) is double-buffered: a sweep reads
// buffer A and writes buffer B (the reference's new_* maps, :328-333), then they swap.
//
// Sharding (one process per GPU): nodes are partitioned by an edge cut (owner[v]).  A rank lays
// out only the nodes it owns.  The pi-message of edge u->v is computed by owner(u), the
// lambda-message by owner(v) (each needs only state stored at that node).  For a CUT edge both
// message halves live in the EXCHANGE region appended to every record buffer:
//
//   [ tile records | segment of rank 0 | segment of rank 1 | ... ]      segment = halves + slots
//
// The layout of the exchange region is identical on every rank and each half has exactly one
// writer (its producer), so one in-place all-gather of the segments per sweep IS the halo
// exchange -- no pack / unpack kernels.  Each segment ends with the rank's 256 residual slots,
// so the same collective also carries max|new-old| to every rank and all ranks stop on the
// same sweep.  With one rank the region holds just the residual slots.
#pragma once

#include <sys/mman.h>

#include <new>
#include <algorithm>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <system_error>
#include <thread>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/bn_mi355x.h"

namespace bnmi {

// Host buffers of tens of MB (the model's flat CPT copy, the lane-striped tile image, the DAG path's padded image): on 4 KiB pages
// a fresh 51 MB vector costs 12 500 first-touch faults -- 30 ms each on the boxes this was measured on, three of them per bn_create of
// the 316 x 316 grid.  Allocations of 4 MiB and more are 2 MiB-aligned and marked MADV_HUGEPAGE (transparent huge pages in `madvise`
// mode: 26 faults instead); smaller ones go to malloc as before.
template <class T>
struct HugeAlloc {
    typedef T value_type;
    HugeAlloc() = default;
    template <class U> HugeAlloc(const HugeAlloc<U>&) {}
    static constexpr size_t kHuge = size_t(2) << 20;
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        void* q = nullptr;
        if (bytes >= 2 * kHuge) {
            const size_t rounded = (bytes + kHuge - 1) & ~(kHuge - 1);
            q = std::aligned_alloc(kHuge, rounded);
            if (q) (void)madvise(q, rounded, MADV_HUGEPAGE);
        } else {
            q = std::malloc(bytes ? bytes : 1);
        }
        if (!q) throw std::bad_alloc();
        return static_cast<T*>(q);
    }
    void deallocate(T* q, size_t) { std::free(q); }
    template <class U> bool operator==(const HugeAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const HugeAlloc<U>&) const { return false; }
};
typedef std::vector<double, HugeAlloc<double>> BigVec;

// Host-side loops over independent nodes / tiles (the CPT images: tens of MB of scattered copies at 10^5 nodes) on a few threads:
// fn(begin, end) over [0, n) in contiguous chunks of at least `grain`; BN_HOST_THREADS caps the count (default: up to 8).
// An exception in a worker (bad_alloc) is rethrown in the caller after all have joined.
template <class F>
inline void parallel_for(int64_t n, int64_t grain, F&& fn) {
    unsigned hw = std::thread::hardware_concurrency();
    int64_t T = std::min<int64_t>(hw ? hw : 1, 8);
    if (const char* t = std::getenv("BN_HOST_THREADS")) T = std::max(1, std::atoi(t));
    T = std::min<int64_t>(T, grain > 0 ? n / grain : n);
    if (T <= 1) { fn(int64_t(0), n); return; }
    std::vector<std::thread> th;
    std::exception_ptr err;
    std::mutex mu;
    const int64_t per = (n + T - 1) / T;
    for (int64_t t = 0; t < T; ++t) {
        const int64_t b = t * per, e = std::min(n, b + per);
        if (b >= e) break;
        auto body = [&, b, e] {
            try { fn(b, e); } catch (...) { std::lock_guard<std::mutex> g(mu); if (!err) err = std::current_exception(); }
        };
        try { th.emplace_back(body); } catch (const std::system_error&) { body(); }   // (no thread to be had: the caller does the chunk)
    }
    for (std::thread& x : th) x.join();
    if (err) std::rethrow_exception(err);
}


constexpr int kWave = 64;
constexpr int kResSlots = 256;              // residual slots per rank (spread the atomics)
constexpr int kResSlotsD2 = kResSlots / 2;  // ... in double2 units (8 B each)

enum Variant : int32_t {
    kVariantGeneric = 0,   // runtime loops, any shape, G = 1
    kVariantUniform = 1,   // templated: node and all parents share k in {2,3,4}, CPT <= 64 entries
    kVariantGroup = 2,     // k = 4, 3..5 parents: G = 4^(m-2) lanes share a node, 64 entries per lane
    kVariantFlat = 3,      // any arities: G = 8..64 lanes per node, entry e of the reference's row-major
                           // CPT in lane e % G of the group; <= 8 parents, arities summing to <= 64
};
constexpr int kFlatMaxParents = 8;
constexpr int kFlowSlotsPerRank = 2048;  // granule slots per rank in a dataflow sync table (= 8 waves x 256 tile blocks)
constexpr int kMaxNbrChunks = 4;   // dataflow form: a tile polls at most 4 x 64 neighbour tiles

// Device-visible shape class.  POD.
struct ClassDesc {
    int32_t kv;                      // own arity
    int32_t m;                       // number of parents
    int32_t kp[BN_MAX_PARENTS];      // parent arities (ascending parent id order)
    int32_t kvp;                     // kv rounded up to even
    int32_t kpp[BN_MAX_PARENTS];     // kp rounded up to even
    int32_t rows;                    // C = prod kp
    int32_t G;                       // lanes per node
    int32_t npt;                     // nodes per tile = 64 / G
    int32_t variant;
    int32_t per_lane;                // CPT entries per lane = kv * rows / G
    int32_t per_lane_pad;            // rounded up to even
    int32_t rec_off[BN_MAX_PARENTS]; // doubles from the tile's rec_base to in-edge j's record
    int32_t rec_doubles;             // per tile
    int32_t cstride[BN_MAX_PARENTS]; // mixed-radix stride of parent j in the assignment index
    int32_t n_nodes;
    // any-arity classes on the ordered path (at most two CPT entries per lane, kv * rows <= 128):
    int32_t flat_tab_off;            // first FlatEntry of the class in Plan::flat_tab (2 G entries), -1: none
    int32_t lam_run[BN_MAX_PARENTS]; // kv * rows / kp[j]: terms one bucket of the lambda-message to parent j sums
    int32_t magic_kv;                // ceil(2^16 / kv): x / kv == (x * magic_kv) >> 16 for 0 <= x < 1024 (kv <= 64)
    int32_t magic_hv;                // ceil(2^20 / (kvp / 2)): chunk stride of a MsgRef, (lam - pi) / (kvp / 2), for differences < 2^11
};

// Per CPT entry e of an ordered any-arity class, everything the tile code would otherwise derive with
// runtime divisions: the digits of e in the mixed radix (own state fastest, last parent next) and the place
// of e's term in each accumulator's summation run (the reference's order: own state outer, assignment inner,
// belief_propagation.hpp:174-200, :240-266).  32 bytes, one pair of 16-byte loads.
struct FlatEntry {
    uint8_t dj[kFlatMaxParents];       // state of parent j in the assignment
    uint8_t pos_lam[kFlatMaxParents];  // lambda-message to parent j: dj[j] * lam_run[j] + own state * (rows / kp[j]) + assignment without digit j
    uint8_t ei;                        // own state
    uint8_t pos_pi;                    // pi(v): ei * rows + assignment
    uint8_t valid;                     // e < kv * rows
    uint8_t pad_[13];
};
static_assert(sizeof(FlatEntry) == 32, "FlatEntry is loaded as two 16-byte words");

// One wavefront of work.  POD, 64 bytes.
struct TileDesc {
    int32_t cls;
    int32_t n_nodes;      // active nodes (<= npt)
    int32_t cmax;         // max out-degree among the tile's nodes
    int32_t slot_base;    // index of lane-slot 0 in per-slot arrays (frozen, slot_boff, ...)
    int64_t cpt_base;     // doubles
    int64_t rec_base;     // doubles
    int64_t node_base;    // doubles: pi at node_base, lambda at node_base + kvp*npt
    int64_t out_base;     // MsgRef entries: out-edge c of node nl at out_base + c*npt + nl
    int64_t in_ref_base;  // -1: in-edge records at rec_base (arithmetic); else MsgRef entries
                          //     [j][nl] -- the tile has a parent on another rank
    // copy of the class fields the kernel dispatches on (saves a dependent load per wave)
    uint8_t kv, m, variant, npt;
    int32_t pad_;
};
static_assert(sizeof(TileDesc) == 64, "TileDesc must stay 64 bytes");

// Where the two messages of one edge live, in double2 units from the start of a record buffer.
//   pi  < 0          : no edge
//   lam >= 0         : tile-resident record; chunk c of the pi-message at pi + c*stride and of
//                      the lambda-message at lam + c*stride, stride = (lam - pi) / H where
//                      H = chunks per message = roundup(k[parent], 2) / 2
//   lam <  0         : cut edge, halves in the exchange region, contiguous chunks:
//                      pi-message at pi + c, lambda-message at ~lam + c
struct MsgRef {
    int32_t pi;
    int32_t lam;
};

struct Plan {
    // model (kept for un-striping / diagnostics)
    int32_t n = 0;
    int64_t E = 0;
    std::vector<int32_t> k, in_ptr, in_idx;
    std::vector<int64_t> node_off;   // [n+1] prefix of k
    std::vector<int64_t> msg_off;    // [E+1] prefix of k[parent]
    // sharding
    int32_t rank = 0, nranks = 1;
    std::vector<int32_t> owner;      // [n] (empty when nranks == 1)
    int32_t n_owned = 0;
    int64_t n_cut_edges = 0;         // cut edges incident to this rank
    // classes / tiles (owned nodes only)
    std::vector<ClassDesc> classes;
    std::vector<FlatEntry> flat_tab;  // per ordered any-arity class, 2 G entries (ClassDesc::flat_tab_off)
    bool latency_rules_applied = false;  // the layout trades wavefront count for the latency of one query (bn_plan.cpp)
    bool wide_requested = false;         // lanes_per_node 3 / 4
    bool group_wide = false;             // the wide lane-group split (16 table entries per lane) is in force
    std::vector<TileDesc> tiles;
    std::vector<int32_t> node_class; // [n]   -1 for nodes of other ranks
    std::vector<int32_t> node_slot;  // [n]   tiles[t].slot_base + nl, or -1
    std::vector<int32_t> node_tile;  // [n]
    std::vector<int32_t> node_nl;    // [n]
    std::vector<int32_t> slot_node;  // [n_slots] node id or -1
    std::vector<int64_t> slot_boff;  // [n_slots] node_off[node] or -1
    int32_t n_slots = 0;
    // per CSR edge: where its messages live on this rank ({-1,0} when neither endpoint is owned)
    std::vector<MsgRef> edge_ref;    // [E]
    // exchange region (double2 units)
    int64_t g_base = 0;              // start of the exchange region in a record buffer
    int64_t seg_d2 = kResSlotsD2;    // segment size per rank, residual slots included
    int64_t seg_data_d2 = 0;         // message halves per segment (max over ranks, padded)
    std::vector<int64_t> seg_used_d2;  // [nranks] halves actually produced by each rank
    // device images
    BigVec cpt_striped;  // released after upload (cpt_doubles keeps the size)
    int64_t cpt_doubles = 0;
    std::vector<int64_t> cpt_off;     // [n+1] reference-order flat CPT (kept for likelihood weighting)
    BigVec cpt_flat;
    std::vector<MsgRef> out_refs;    // per tile [c][nl]
    std::vector<MsgRef> in_refs;     // per boundary tile [j][nl]
    int64_t rec_doubles = 0;         // tile records only
    int64_t rec_total_doubles = 0;   // tile records + exchange region (one buffer)
    int64_t node_doubles = 0;
    // metrics (this rank's share)
    int64_t algorithmic_bytes = 0, layout_bytes = 0, messages_per_sweep = 0;
    int32_t g_max = 1;
    int32_t n_interior_tiles = 0;    // tiles [0, n_interior_tiles) touch no cut edge (all tiles when nranks == 1)
    // Dataflow form of the resident kernel (bn_resident.hip): a tile waits for ITS neighbours only -- the tiles that
    // hold a parent or a child of one of its nodes, i.e. every tile that writes a message it reads or reads one it
    // writes -- instead of for a grid barrier.  Neighbour t of tile T at nbr[(T * nbr_chunks + t / kWave) * kWave +
    // t % kWave] (lane t % kWave polls it in round t / kWave), -1 beyond the tile's count.  Empty when some tile has
    // more than kMaxNbrChunks * kWave neighbours (the plan is then not eligible for the dataflow form).
    std::vector<int32_t> nbr;        // [n_tiles * nbr_chunks * kWave]
    int32_t nbr_max = 0;             // most neighbours any tile has
    int32_t nbr_chunks = 1;          // ceil(nbr_max / kWave)
    // Sharded plans: the neighbour tiles on THIS rank (CSR over tiles) and, per cut edge incident to this rank, the
    // local tile, the peer rank and the peer's node; the peer's tile of that node arrives with its export blob
    // (bn_peer_import), which turns the two into the table above with slots rank * kFlowSlotsPerRank + tile.
    std::vector<int32_t> nbl_ptr, nbl_idx;
    struct CutLink { int32_t tile, rank, node; };
    std::vector<CutLink> cut_links;
    std::vector<int32_t> boundary_node, boundary_tile;  // owned nodes with a cut edge (ascending) and their tiles
    int32_t variants = 0;            // bit v set: some class is of Variant v (selects the kernel instantiation)
    bool light = false;              // no register-resident / k = 4 lane-group tile: the high-occupancy launch applies
};

struct ShardSpec {
    int32_t rank = 0;
    int32_t nranks = 1;
    const int32_t* owner = nullptr;  // [n]; nullptr with nranks > 1 -> balanced contiguous ranges
};

// Balanced contiguous node ranges by CPT bytes (grid, row-major ids -> row stripes).
void default_owner(const bn_model_desc& d, int32_t nranks, std::vector<int32_t>& owner);

// Builds the plan.  Returns empty string on success, else an error message (BN_ERR_ARG).
std::string build_plan(const bn_model_desc& d, const ShardSpec& shard, Plan& out);

// p.cpt_striped (p.cpt_doubles doubles) from a flat CPT array in the model's layout; build_plan and bn_reload_cpt
void stripe_cpt(Plan& p, const double* cpt);

// (Re)builds Plan::nbr / nbr_max / nbr_chunks: local neighbour tiles (nbl_*) + the given slots of neighbour tiles on
// other ranks, per tile.  Returns an error text when a tile has too many neighbours (nbr is then empty).
std::string build_neighbour_table(Plan& p, const std::vector<std::vector<int32_t>>& remote_slots);

// Host un-striping of a record buffer into CSR edge order (diagnostics, bn_bp_messages).
// Edges without an owned endpoint are left untouched.
void unstripe_messages(const Plan& p, const std::vector<double>& rec, double* pi_msg, double* lambda_msg);

}  // namespace bnmi

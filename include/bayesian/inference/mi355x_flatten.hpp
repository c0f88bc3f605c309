// mi355x_flatten.hpp -- graph_t / cpt_t  ->  flat model of bn_mi355x.h, through the PUBLIC
// interface of the data model only (works with the reference's bayesian/graph.hpp as well as
// with include/compat/bayesian/graph.hpp).
//
// Contract reproduced from the reference:
//   node id      = position in graph_t::vertex_list()            (graph.hpp:214), never vertex_t::id
//   parent order = graph_t::in_vertexes(v), i.e. ascending position (graph.hpp:389-413)
//   CPT row      = mixed radix over the parents, FIRST parent most significant -- the order
//                  all_combination_pattern enumerates (belief_propagation.hpp:269-295)
// A missing CPT row, which the reference dereferences as a dangling vector (graph.hpp:120-124),
// is reported as std::runtime_error here.
#ifndef BN_MI355X_FLATTEN_HPP
#define BN_MI355X_FLATTEN_HPP

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

#include <bayesian/graph.hpp>
#include <bayesian/matrix.hpp>

#include "../../bn_mi355x.h"

namespace bn {
namespace mi355x {

// vertex -> position in vertex_list(), for the per-query marshalling of evidence: an open-addressing table keyed by the
// vertex's address (one cache line per lookup, and the line can be requested ahead: prefetch()), beside flat_model::index
// (a node-based std::unordered_map: two to three dependent misses per lookup on a 10^5-node network).
class position_table {
public:
    void build(std::vector<vertex_type> const& nodes)
    {
        unsigned bits = 4;
        while((std::size_t(1) << bits) < 2 * nodes.size() + 2) ++bits;
        shift_ = 64 - bits;
        slots_.assign(std::size_t(1) << bits, slot{nullptr, -1});
        for(std::size_t i = 0; i < nodes.size(); ++i)
        {
            std::size_t s = home(nodes[i].get());
            while(slots_[s].key != nullptr && slots_[s].key != nodes[i].get()) s = (s + 1) & (slots_.size() - 1);
            slots_[s] = slot{nodes[i].get(), static_cast<std::int32_t>(i)};
        }
    }
    void prefetch(void const* vertex) const
    {
#if defined(__GNUC__)
        __builtin_prefetch(&slots_[home(vertex)]);
#else
        (void)vertex;
#endif
    }
    // -1: not a vertex of the graph
    std::int32_t find(void const* vertex) const
    {
        if(slots_.empty()) return -1;
        for(std::size_t s = home(vertex);; s = (s + 1) & (slots_.size() - 1))
        {
            if(slots_[s].key == vertex) return slots_[s].position;
            if(slots_[s].key == nullptr) return -1;
        }
    }

private:
    struct slot { void const* key; std::int32_t position; };
    std::size_t home(void const* vertex) const
    {
        return static_cast<std::size_t>((static_cast<std::uint64_t>(reinterpret_cast<std::uintptr_t>(vertex)) * 0x9E3779B97F4A7C15ull) >> shift_);
    }
    std::vector<slot> slots_;
    unsigned shift_ = 60;
};

struct flat_model {
    std::vector<vertex_type> nodes;                    // index -> vertex (copy of vertex_list())
    std::unordered_map<vertex_type, std::int32_t> index;  // vertex -> index
    position_table lookup;                             // the same mapping, for the per-query path
    std::vector<std::int32_t> k, in_ptr, in_idx;
    std::vector<std::int64_t> cpt_off, node_off;
    std::vector<double> cpt;

    bn_model_desc desc(int device = BN_DEVICE_CURRENT) const
    {
        bn_model_desc d;
        d.n_nodes = static_cast<std::int32_t>(k.size());
        d.k = k.data();
        d.in_ptr = in_ptr.data();
        d.in_idx = in_idx.data();
        d.cpt_off = cpt_off.data();
        d.cpt = cpt.data();
        d.device = device;
        d.lanes_per_node = 0;
        return d;
    }
};

// fn(begin, end) over [0, n) in contiguous chunks on a few threads (BN_HOST_THREADS caps the count; default up to 32, one
// thread below `grain` items per thread).  An exception in a worker is rethrown in the caller after all have joined.
template <class F>
inline void for_chunks(std::size_t const n, std::size_t const grain, F&& fn)
{
    std::size_t threads = std::min<std::size_t>(std::max(1u, std::thread::hardware_concurrency()), 32);
    if(char const* t = std::getenv("BN_HOST_THREADS")) threads = static_cast<std::size_t>(std::max(1, std::atoi(t)));
    threads = std::min(threads, grain ? n / grain : n);
    if(threads <= 1) { fn(std::size_t(0), n); return; }
    std::vector<std::thread> pool;
    std::exception_ptr error;
    std::mutex guard;
    std::size_t const per = (n + threads - 1) / threads;
    for(std::size_t t = 0; t < threads; ++t)
    {
        std::size_t const b = t * per, e = std::min(n, b + per);
        if(b >= e) break;
        auto body = [&, b, e]
        {
            try { fn(b, e); }
            catch(...) { std::lock_guard<std::mutex> lock(guard); if(!error) error = std::current_exception(); }
        };
        try { pool.emplace_back(body); }
        catch(std::system_error const&) { body(); }   // no thread to be had: this one does the chunk
    }
    for(auto& th : pool) th.join();
    if(error) std::rethrow_exception(error);
}

// Structure only (arities, parents, CPT offsets); fm.cpt stays empty.  with_cpt = true also reads
// every CPT row.
// Reading the rows is what a functor's construction costs: cpt_t's interface hands out ONE row per call, keyed by a condition_t (an
// unordered_map) that operator[] takes BY VALUE (graph.hpp:108) -- a map copy, its hash and a find per row, 1.6 M times on the
// 316 x 316 grid (0.3 s on one core).  The rows of different nodes are independent and the graph is only read: nodes are spread
// over a few threads, and a node's condition is kept and stepped like an odometer instead of being rebuilt per row.
inline flat_model flatten_impl(graph_t const& graph, bool const with_cpt)
{
    flat_model fm;
    fm.nodes = graph.vertex_list();
    std::size_t const n = fm.nodes.size();
    fm.index.reserve(n);
    for(std::size_t i = 0; i < n; ++i) fm.index[fm.nodes[i]] = static_cast<std::int32_t>(i);
    fm.lookup.build(fm.nodes);

    fm.k.resize(n);
    fm.in_ptr.assign(n + 1, 0);
    fm.cpt_off.assign(n + 1, 0);
    fm.node_off.assign(n + 1, 0);
    for(std::size_t i = 0; i < n; ++i)
    {
        fm.k[i] = static_cast<std::int32_t>(fm.nodes[i]->selectable_num);
        fm.node_off[i + 1] = fm.node_off[i] + fm.k[i];
    }
    for(std::size_t i = 0; i < n; ++i)
    {
        auto const parents = graph.in_vertexes(fm.nodes[i]);
        std::size_t rows = 1;
        for(auto const& p : parents)
        {
            auto const it = fm.index.find(p);
            if(it == fm.index.end()) throw std::runtime_error("bn::mi355x::flatten: parent is not in vertex_list()");
            fm.in_idx.push_back(it->second);
            rows *= static_cast<std::size_t>(fm.k[it->second]);
        }
        fm.in_ptr[i + 1] = static_cast<std::int32_t>(fm.in_idx.size());
        fm.cpt_off[i + 1] = fm.cpt_off[i] + static_cast<std::int64_t>(rows) * fm.k[i];
    }
    if(!with_cpt) return fm;

    fm.cpt.resize(static_cast<std::size_t>(fm.cpt_off[n]));
    for_chunks(n, 256, [&](std::size_t const begin, std::size_t const end)
    {
        for(std::size_t i = begin; i < end; ++i)
        {
            std::size_t const m = static_cast<std::size_t>(fm.in_ptr[i + 1] - fm.in_ptr[i]);
            std::int32_t const* const par = fm.in_idx.data() + fm.in_ptr[i];
            std::size_t const kv = static_cast<std::size_t>(fm.k[i]);
            // every parent assignment, first parent slowest
            std::vector<int> state(m, 0);
            condition_t cond;
            for(std::size_t j = 0; j < m; ++j) cond[fm.nodes[par[j]]] = 0;
            double* out = fm.cpt.data() + fm.cpt_off[i];
            for(std::int64_t o = fm.cpt_off[i]; o < fm.cpt_off[i + 1]; o += fm.k[i], out += kv)
            {
                auto const entry = fm.nodes[i]->cpt[cond];
                if(!entry.first || entry.second.size() != kv)
                    throw std::runtime_error("bn::mi355x::flatten: CPT row missing or of wrong length at node " + std::to_string(i));
                std::copy(entry.second.begin(), entry.second.end(), out);
                for(std::size_t j = m; j-- > 0;)   // the next assignment: only the digits that change are written
                {
                    if(++state[j] < fm.k[par[j]]) { cond[fm.nodes[par[j]]] = state[j]; break; }
                    state[j] = 0;
                    cond[fm.nodes[par[j]]] = 0;
                }
            }
        }
    });
    return fm;
}

inline flat_model flatten(graph_t const& graph) { return flatten_impl(graph, true); }
inline flat_model flatten_structure(graph_t const& graph) { return flatten_impl(graph, false); }

// Inverse of flatten for the CPTs: write the flat rows back into every node's cpt_t (which must
// already hold one row per parent assignment, i.e. cpt.assign(parents, node) was called).
inline void store_cpts(graph_t const& graph, flat_model const& fm, std::vector<double> const& cpt)
{
    for(std::size_t i = 0; i < fm.nodes.size(); ++i)
    {
        auto const parents = graph.in_vertexes(fm.nodes[i]);
        std::vector<int> state(parents.size(), 0);
        std::size_t const k = static_cast<std::size_t>(fm.k[i]);
        for(std::int64_t o = fm.cpt_off[i]; o < fm.cpt_off[i + 1]; o += fm.k[i])
        {
            condition_t cond;
            for(std::size_t j = 0; j < parents.size(); ++j) cond[parents[j]] = state[j];
            auto entry = fm.nodes[i]->cpt[cond];
            if(!entry.first) throw std::runtime_error("bn::mi355x::store_cpts: CPT row missing at node " + std::to_string(i));
            entry.second.assign(cpt.begin() + o, cpt.begin() + o + static_cast<std::int64_t>(k));
            for(std::size_t j = parents.size(); j-- > 0;)
            {
                if(++state[j] < fm.k[fm.in_idx[fm.in_ptr[i] + j]]) break;
                state[j] = 0;
            }
        }
    }
}

// RAII over the C handle; non-zero codes become std::runtime_error (no exception crosses the ABI).
// Move-only: a functor that must be copyable (belief_propagation, like the reference's) builds a
// second engine from its own flat model, so the copies stay as independent as the reference's.
class engine_handle {
public:
    engine_handle() = default;
    explicit engine_handle(flat_model const& fm, int device = BN_DEVICE_CURRENT)
    {
        bn_model_desc const d = fm.desc(device);
        check(bn_create(&d, &handle_));
    }
    engine_handle(engine_handle const&) = delete;
    engine_handle& operator=(engine_handle const&) = delete;
    engine_handle(engine_handle&& other) noexcept : handle_(other.handle_) { other.handle_ = nullptr; }
    engine_handle& operator=(engine_handle&& other) noexcept
    {
        if(this != &other)
        {
            if(handle_) bn_destroy(handle_);
            handle_ = other.handle_;
            other.handle_ = nullptr;
        }
        return *this;
    }
    ~engine_handle() { if(handle_) bn_destroy(handle_); }

    bn_engine* get() const { return handle_; }
    static void check(int rc)
    {
        if(rc < 0) throw std::runtime_error(std::string("bn_mi355x: ") + bn_last_error());
    }

private:
    bn_engine* handle_ = nullptr;
};

// New CPT values on an unchanged structure -> the engine (bn_reload_cpt).  The reference's functors read node->cpt at every call
// (belief_propagation.hpp:61,186,252; likelihood_weighting.hpp:148-158), the engines hold device images made when the functor was
// built: a functor's reload() brings them up to date.  Throws when vertices, arities or edges are no longer the ones flattened.
inline void reload_cpts(graph_t const& graph, flat_model& fm, engine_handle& engine)
{
    flat_model now = flatten(graph);
    if(now.nodes != fm.nodes || now.k != fm.k || now.in_ptr != fm.in_ptr || now.in_idx != fm.in_idx)
        throw std::runtime_error("bn::mi355x::reload_cpts: the graph's structure changed since the functor was built (build a new one)");
    engine_handle::check(bn_reload_cpt(engine.get(), now.cpt.data(), static_cast<std::int64_t>(now.cpt.size())));
    fm.cpt.swap(now.cpt);
}

} // namespace mi355x
} // namespace bn

#endif // BN_MI355X_FLATTEN_HPP

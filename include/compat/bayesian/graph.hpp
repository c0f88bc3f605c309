// compat/bayesian/graph.hpp -- this repository's own, API-compatible stand-in for the reference's
// data model (bayesian/graph.hpp: vertex_t, cpt_t, graph_t), for building and testing the
// drop-in inference headers where the reference is not installed (the GPU box), and for graphs
// the reference cannot hold: its graph_t is a dense V x V matrix of shared_ptr (graph.hpp:485,
// 160 GB at 10^5 nodes) with O(V^2) neighbour lookups; this one keeps sorted adjacency lists.
// Same public names and semantics (neighbours enumerate in ascending vertex_list() position,
// add_edge refuses duplicates and cycles); the implementation is independent.
#ifndef BNI_GRAPH_HPP
#define BNI_GRAPH_HPP

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace bn {

struct vertex_t;
struct edge_t;
typedef std::shared_ptr<vertex_t> vertex_type;
typedef std::shared_ptr<edge_t> edge_type;
typedef std::unordered_map<vertex_type, int> condition_t;

} // namespace bn

namespace std {
// order-independent hash of a parent assignment, so that condition_t can key a table
template<> struct hash<bn::condition_t> {
    std::size_t operator()(bn::condition_t const& cond) const noexcept
    {
        std::size_t h = 0x9e3779b97f4a7c15ull ^ cond.size();
        for(auto const& kv : cond)
        {
            std::size_t const a = std::hash<bn::vertex_type>()(kv.first);
            std::size_t const b = std::hash<int>()(kv.second);
            h += (a * 0x100000001b3ull) ^ (b + 0x9e3779b9u + (a << 6) + (a >> 2));
        }
        return h;
    }
};
} // namespace std

namespace bn {

// Conditional probability table: parent assignment -> row of P(target | parents)
class cpt_t {
public:
    typedef std::unordered_map<condition_t, std::vector<double>> table_type;

    cpt_t() = default;
    cpt_t(std::vector<vertex_type> const& parent_nodes, vertex_type const& target_node)
    {
        assign(parent_nodes, target_node);
    }

    // allocate one zero row per parent assignment
    inline void assign(std::vector<vertex_type> const& parent_nodes, vertex_type const& target_node);

    // rows whose assignment agrees with `cond` on every variable of `cond`
    table_type filter(condition_t const& cond)
    {
        table_type out;
        for(auto const& row : table_)
        {
            bool match = true;
            for(auto const& want : cond)
            {
                auto const it = row.first.find(want.first);
                if(it == row.first.end() || it->second != want.second) { match = false; break; }
            }
            if(match) out.insert(row);
        }
        return out;
    }

    std::vector<vertex_type> condition_node() { return parents_; }

    std::vector<condition_t> pattern()
    {
        std::vector<condition_t> out;
        out.reserve(table_.size());
        for(auto const& row : table_) out.push_back(row.first);
        return out;
    }

    // (found, row).  When not found the row is an empty scratch vector.
    std::pair<bool, std::vector<double>&> operator[](condition_t const cond)
    {
        auto const it = table_.find(cond);
        if(it == table_.end()) { missing_.clear(); return std::pair<bool, std::vector<double>&>(false, missing_); }
        return std::pair<bool, std::vector<double>&>(true, it->second);
    }
    std::pair<bool, std::vector<double> const&> operator[](condition_t const cond) const
    {
        auto const it = table_.find(cond);
        if(it == table_.end()) return std::pair<bool, std::vector<double> const&>(false, missing_);
        return std::pair<bool, std::vector<double> const&>(true, it->second);
    }

private:
    std::vector<vertex_type> parents_;
    table_type table_;
    std::vector<double> missing_;
};

struct vertex_t {
    int id = 0;
    std::size_t selectable_num = 0;
    cpt_t cpt;
};

struct edge_t {};

struct database_t {
    std::string graph_name;
    std::unordered_map<std::size_t, std::string> node_name;
    std::unordered_map<std::size_t, std::vector<std::string>> options_name;
};

inline void cpt_t::assign(std::vector<vertex_type> const& parent_nodes, vertex_type const& target_node)
{
    table_type fresh;
    std::vector<std::size_t> state(parent_nodes.size(), 0);
    bool done = false;
    while(!done)
    {
        condition_t cond;
        for(std::size_t j = 0; j < parent_nodes.size(); ++j) cond[parent_nodes[j]] = static_cast<int>(state[j]);
        fresh.emplace(cond, std::vector<double>(target_node->selectable_num));
        done = true;
        for(std::size_t j = parent_nodes.size(); j-- > 0;)
        {
            if(++state[j] < parent_nodes[j]->selectable_num) { done = false; break; }
            state[j] = 0;
        }
    }
    parents_ = parent_nodes;
    table_.swap(fresh);
}

// Directed acyclic graph over shared vertices.
class graph_t {
public:
    graph_t() = default;
    virtual ~graph_t() = default;
    graph_t(graph_t const&) = default;
    graph_t(graph_t&& other) { other.swap(*this); }
    graph_t& operator=(graph_t const& rhs) { graph_t(rhs).swap(*this); return *this; }
    graph_t& operator=(graph_t&& rhs) { rhs.swap(*this); return *this; }

    void swap(graph_t& other) noexcept
    {
        vertices_.swap(other.vertices_);
        edges_.swap(other.edges_);
        out_.swap(other.out_);
        in_.swap(other.in_);
        position_.swap(other.position_);
        ends_.swap(other.ends_);
    }

    std::vector<vertex_type> const& vertex_list() const { return vertices_; }
    std::vector<edge_type> const& edge_list() const { return edges_; }

    graph_t clone() const
    {
        graph_t g;
        for(auto const& v : vertices_) *g.add_vertex() = *v;
        for(auto const& e : edges_)
        {
            auto const& ab = ends_.at(e.get());
            g.add_edge(g.vertices_[ab.first], g.vertices_[ab.second]);
        }
        return g;
    }

    vertex_type add_vertex()
    {
        auto v = std::make_shared<vertex_t>();
        position_[v.get()] = vertices_.size();
        vertices_.push_back(v);
        out_.emplace_back();
        in_.emplace_back();
        return v;
    }

    // nullptr when an endpoint is unknown, the edge exists, or it would close a cycle
    edge_type add_edge(vertex_type const& from, vertex_type const& to)
    {
        std::size_t const a = index_of(from), b = index_of(to);
        if(a == npos || b == npos) return nullptr;
        if(is_able_trace(to, from)) return nullptr;
        if(find_slot(out_[a], b) != out_[a].end() && find_slot(out_[a], b)->first == b) return nullptr;
        auto e = std::make_shared<edge_t>();
        edges_.push_back(e);
        out_[a].insert(find_slot(out_[a], b), std::make_pair(b, e));
        in_[b].insert(find_slot(in_[b], a), std::make_pair(a, e));
        ends_[e.get()] = std::make_pair(a, b);
        return e;
    }

    bool erase_vertex(vertex_type const& v)
    {
        std::size_t const x = index_of(v);
        if(x == npos) return false;
        std::vector<edge_type> doomed;
        for(auto const& s : out_[x]) doomed.push_back(s.second);
        for(auto const& s : in_[x]) doomed.push_back(s.second);
        for(auto const& e : doomed) erase_edge(e);
        vertices_.erase(vertices_.begin() + x);
        out_.erase(out_.begin() + x);
        in_.erase(in_.begin() + x);
        reindex_after(x);
        return true;
    }

    bool erase_edge(edge_type const& e)
    {
        auto const it = ends_.find(e.get());
        if(it == ends_.end()) return false;
        std::size_t const a = it->second.first, b = it->second.second;
        out_[a].erase(find_slot(out_[a], b));
        in_[b].erase(find_slot(in_[b], a));
        ends_.erase(it);
        edges_.erase(std::remove(edges_.begin(), edges_.end(), e), edges_.end());
        return true;
    }

    bool erase_all_vertex()
    {
        graph_t().swap(*this);
        return true;
    }

    bool erase_all_edge()
    {
        edges_.clear();
        ends_.clear();
        for(auto& l : out_) l.clear();
        for(auto& l : in_) l.clear();
        return true;
    }

    // reverse an edge; on failure the original edge is put back and nullptr returned
    edge_type change_edge_direction(edge_type const& e)
    {
        auto const it = ends_.find(e.get());
        if(it == ends_.end()) return nullptr;
        vertex_type const from = vertices_[it->second.first], to = vertices_[it->second.second];
        erase_edge(e);
        if(auto const flipped = add_edge(to, from)) return flipped;
        add_edge(from, to);
        return nullptr;
    }

    std::vector<edge_type> out_edges(vertex_type const& from) const { return edges_of(out_, from); }
    std::vector<edge_type> in_edges(vertex_type const& to) const { return edges_of(in_, to); }
    std::vector<vertex_type> out_vertexes(vertex_type const& from) const { return neighbours_of(out_, from); }
    std::vector<vertex_type> in_vertexes(vertex_type const& to) const { return neighbours_of(in_, to); }

    vertex_type source(edge_type const& e) const
    {
        auto const it = ends_.find(e.get());
        return it == ends_.end() ? nullptr : vertices_[it->second.first];
    }
    vertex_type target(edge_type const& e) const
    {
        auto const it = ends_.find(e.get());
        return it == ends_.end() ? nullptr : vertices_[it->second.second];
    }

    // is `to` reachable from `from` along edge directions (a vertex reaches itself)
    bool is_able_trace(vertex_type const& from, vertex_type const& to) const
    {
        std::size_t const a = index_of(from), b = index_of(to);
        if(a == npos || b == npos) return false;
        if(a == b) return true;
        std::vector<char> seen(vertices_.size(), 0);
        std::vector<std::size_t> stack(1, a);
        seen[a] = 1;
        while(!stack.empty())
        {
            std::size_t const x = stack.back();
            stack.pop_back();
            for(auto const& s : out_[x])
            {
                if(s.first == b) return true;
                if(!seen[s.first]) { seen[s.first] = 1; stack.push_back(s.first); }
            }
        }
        return false;
    }

private:
    typedef std::pair<std::size_t, edge_type> slot_t;     // (neighbour position, edge), sorted by position
    typedef std::vector<slot_t> slots_t;
    static constexpr std::size_t npos = std::numeric_limits<std::size_t>::max();

    std::size_t index_of(vertex_type const& v) const
    {
        auto const it = position_.find(v.get());
        return it == position_.end() ? npos : it->second;
    }
    static slots_t::iterator find_slot(slots_t& l, std::size_t key)
    {
        return std::lower_bound(l.begin(), l.end(), key, [](slot_t const& s, std::size_t k) { return s.first < k; });
    }
    std::vector<edge_type> edges_of(std::vector<slots_t> const& adj, vertex_type const& v) const
    {
        std::vector<edge_type> out;
        std::size_t const x = index_of(v);
        if(x != npos) for(auto const& s : adj[x]) out.push_back(s.second);
        return out;
    }
    std::vector<vertex_type> neighbours_of(std::vector<slots_t> const& adj, vertex_type const& v) const
    {
        std::vector<vertex_type> out;
        std::size_t const x = index_of(v);
        if(x != npos) for(auto const& s : adj[x]) out.push_back(vertices_[s.first]);
        return out;
    }
    void reindex_after(std::size_t removed)
    {
        position_.clear();
        for(std::size_t i = 0; i < vertices_.size(); ++i) position_[vertices_[i].get()] = i;
        for(auto& l : out_) for(auto& s : l) if(s.first > removed) --s.first;
        for(auto& l : in_) for(auto& s : l) if(s.first > removed) --s.first;
        for(auto& kv : ends_)
        {
            if(kv.second.first > removed) --kv.second.first;
            if(kv.second.second > removed) --kv.second.second;
        }
    }

    std::vector<vertex_type> vertices_;
    std::vector<edge_type> edges_;
    std::vector<slots_t> out_, in_;
    std::unordered_map<vertex_t const*, std::size_t> position_;
    std::unordered_map<edge_t const*, std::pair<std::size_t, std::size_t>> ends_;
};

inline void swap(graph_t& lhs, graph_t& rhs) noexcept { lhs.swap(rhs); }

} // namespace bn

#endif // BNI_GRAPH_HPP

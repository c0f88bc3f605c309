// mi355x_marginals.hpp -- the marginals of a run WITHOUT the reference's return type.
//
// The reference returns std::unordered_map<vertex_type, matrix_type> (belief_propagation.hpp:14,
// :151-158; likelihood_weighting.hpp:18; rejection_sampling.hpp:16): per node a hash node, a
// vector<vector<double>> and its one row -- three heap blocks each, built per call and torn down
// by the caller.  On the 99 856-node grid of BASELINE configs[2] that is ~10 ms to build and ~5 ms
// to destroy around a GPU run of 0.14 ms (INTEGRATION.md section 1 has the table).  The functors'
// operator() keeps that type (it is the drop-in contract); their run() returns this view instead:
// the engine's flat node-major array read in place, nothing allocated, valid until the next call
// on the same functor (or its destruction).
//
//     bn::inference::belief_propagation bp(graph);
//     auto const m = bp.run(precondition, epsilon);          // marginals_view
//     double const* p = m[vertex];                           // k(vertex) probabilities
//     for(auto const& e : m) use(e.vertex, e.p, e.k);        // vertex_list() order
//     auto map = m.to_map();                                 // what operator() returns
#ifndef BN_MI355X_MARGINALS_HPP
#define BN_MI355X_MARGINALS_HPP

#include <cstddef>
#include <iterator>
#include <stdexcept>
#include <tuple>
#include <unordered_map>
#include <utility>

#include "mi355x_flatten.hpp"

namespace bn {
namespace mi355x {

class marginals_view {
public:
    struct entry {
        vertex_type const& vertex;
        double const* p;       // k probabilities of `vertex`
        std::size_t k;
        double const* begin() const { return p; }
        double const* end() const { return p + k; }
        double operator[](std::size_t const state) const { return p[state]; }
    };

    class const_iterator {
    public:
        typedef std::forward_iterator_tag iterator_category;
        typedef entry value_type;
        typedef std::ptrdiff_t difference_type;
        typedef entry const* pointer;
        typedef entry reference;

        const_iterator(marginals_view const* view, std::size_t const i) : view_(view), i_(i) {}
        entry operator*() const { return view_->at_position(i_); }
        const_iterator& operator++() { ++i_; return *this; }
        const_iterator operator++(int) { const_iterator old(*this); ++i_; return old; }
        bool operator==(const_iterator const& rhs) const { return i_ == rhs.i_; }
        bool operator!=(const_iterator const& rhs) const { return i_ != rhs.i_; }

    private:
        marginals_view const* view_;
        std::size_t i_;
    };

    marginals_view() = default;
    marginals_view(flat_model const& model, double const* data) : model_(&model), data_(data) {}

    std::size_t size() const { return model_ ? model_->nodes.size() : 0; }   // nodes
    bool empty() const { return size() == 0; }
    const_iterator begin() const { return const_iterator(this, 0); }
    const_iterator end() const { return const_iterator(this, size()); }

    // by vertex, like return_type::at (std::out_of_range for a vertex that is not in the graph)
    double const* operator[](vertex_type const& v) const { return data_ + model_->node_off[position(v)]; }
    double const* at(vertex_type const& v) const { return operator[](v); }
    std::size_t k(vertex_type const& v) const { return static_cast<std::size_t>(model_->k[position(v)]); }
    bool count(vertex_type const& v) const { return model_ && model_->lookup.find(v.get()) >= 0; }
    // a copy in the reference's cell type: what return_type::at(v) holds
    matrix_type matrix(vertex_type const& v) const
    {
        std::size_t const i = position(v);
        matrix_type m(1, static_cast<std::size_t>(model_->k[i]));
        m.assign(data_ + model_->node_off[i], data_ + model_->node_off[i + 1]);
        return m;
    }

    // by position in graph_t::vertex_list() (the node id of bn_mi355x.h)
    entry at_position(std::size_t const i) const
    {
        return entry{model_->nodes[i], data_ + model_->node_off[i], static_cast<std::size_t>(model_->k[i])};
    }
    // the flat array itself: node-major, node_off()[i] .. node_off()[i + 1] are node i's probabilities
    double const* data() const { return data_; }
    std::size_t doubles() const { return model_ ? static_cast<std::size_t>(model_->node_off.back()) : 0; }
    std::vector<std::int64_t> const& node_off() const { return model_->node_off; }

    // The reference's return type.  Every entry is constructed in place -- hash node, row table, row: the three heap
    // blocks the type itself needs (`matrix_type m...; result[v] = m` costs five: a default-constructed entry plus a
    // copy assignment; neither the reference's matrix_type nor the stand-in has a move constructor, both declare a
    // virtual destructor).
    template<class Map = std::unordered_map<vertex_type, matrix_type>>
    Map to_map() const
    {
        Map result;
        std::size_t const n = size();
        result.reserve(n);
        for(std::size_t i = 0; i < n; ++i)
        {
            std::size_t const kv = static_cast<std::size_t>(model_->k[i]);
            auto const it = result.emplace(std::piecewise_construct, std::forward_as_tuple(model_->nodes[i]),
                                           std::forward_as_tuple(std::size_t(1), kv)).first;
            double const* const src = data_ + model_->node_off[i];
            std::vector<double>& row = it->second[0];
            for(std::size_t j = 0; j < kv; ++j) row[j] = src[j];
        }
        return result;
    }

private:
    std::size_t position(vertex_type const& v) const
    {
        if(!model_) throw std::out_of_range("bn::mi355x::marginals_view: empty view");
        std::int32_t const i = model_->lookup.find(v.get());
        if(i < 0) throw std::out_of_range("bn::mi355x::marginals_view: vertex is not in the graph");
        return static_cast<std::size_t>(i);
    }

    flat_model const* model_ = nullptr;
    double const* data_ = nullptr;
};

} // namespace mi355x
} // namespace bn

#endif // BN_MI355X_MARGINALS_HPP

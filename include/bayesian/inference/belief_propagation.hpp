// bayesian/inference/belief_propagation.hpp -- MI355X drop-in for the reference header of the
// same path.  Same class, same public surface (reference belief_propagation.hpp:12-31):
//
//     bn::inference::belief_propagation bp(graph);
//     auto marginals = bp(precondition, epsilon);      // epsilon defaults to 0.001
//     auto marginals = bp(epsilon);                    // no evidence
//
// Put this repository's include/ in front of the reference's include path; graph.hpp and
// matrix.hpp stay the user's.  The functor flattens the graph once (constructor), keeps the
// network resident on the GPU, and every call runs the HIP kernels behind bn_mi355x.h.
// Header-only C++14; link with -lbn_mi355x.
#ifndef BNI_INFERENCE_BELIEF_PROPAGATION_HPP
#define BNI_INFERENCE_BELIEF_PROPAGATION_HPP

#include <algorithm>
#include <unordered_map>
#include <vector>

#include "mi355x_flatten.hpp"
#include "mi355x_marginals.hpp"

namespace bn {
namespace inference {

class belief_propagation {
public:
    typedef std::unordered_map<vertex_type, matrix_type> return_type;

    // The reference copies the graph (graph_t const graph_); vertices are shared_ptrs, so the
    // keys of the returned map are the caller's vertices either way.
    explicit belief_propagation(graph_t const& graph)
        : graph_(graph), model_(mi355x::flatten(graph)), engine_(model_)
    {
    }

    // Copyable like the reference's functor (implicit copy of graph_ and the scratch maps,
    // belief_propagation.hpp:12-21, :320-333): the copy owns a second device engine built from the
    // same flat model, so two copies never share mutable state.
    belief_propagation(belief_propagation const& other)
        : graph_(other.graph_), model_(other.model_), engine_(model_), last_sweeps_(other.last_sweeps_), last_residual_(other.last_residual_)
    {
    }
    belief_propagation& operator=(belief_propagation const& other)
    {
        if(this != &other)
        {
            graph_ = other.graph_;
            model_ = other.model_;
            engine_ = mi355x::engine_handle(model_);
            last_sweeps_ = other.last_sweeps_;
            last_residual_ = other.last_residual_;
        }
        return *this;
    }
    belief_propagation(belief_propagation&&) = default;
    belief_propagation& operator=(belief_propagation&&) = default;

    virtual ~belief_propagation() = default;

    // By-pass (reference :24-28)
    inline return_type operator()(double const epsilon = 0.001)
    {
        std::unordered_map<vertex_type, matrix_type> const precondition;
        return operator()(precondition, epsilon);
    }

    // Run: Loopy Belief Propagation (reference :31-159).  Each evidence entry is a 1 x k matrix
    // that becomes both pi and lambda of its node (:68-73).  The reference's return type costs three heap
    // blocks per node to build and as many for the caller to free -- two orders of magnitude above the run
    // itself on a 10^5-node network; run() below is the same query without it.
    return_type operator()(std::unordered_map<vertex_type, matrix_type> const& precondition, double const epsilon = 0.001)
    {
        return run(precondition, epsilon).to_map();
    }

    // Not in the reference: the same run, the marginals read in place (mi355x_marginals.hpp) -- a non-owning view
    // of the engine's page-locked result buffer, valid until the next call on this functor.  operator() is
    // run(...).to_map(): same bits.
    typedef mi355x::marginals_view view_type;
    inline view_type run(double const epsilon = 0.001)
    {
        std::unordered_map<vertex_type, matrix_type> const precondition;
        return run(precondition, epsilon);
    }
    view_type run(std::unordered_map<vertex_type, matrix_type> const& precondition, double const epsilon = 0.001)
    {
        marshal(precondition, scratch_);
        return run(scratch_, epsilon);
    }

    // An evidence set in the form the C ABI takes (bn_bp_run: positions in vertex_list(), offsets, the 1 x k vectors end to
    // end).  A caller that asks the same query again -- or builds its evidence as arrays in the first place -- prepares it once
    // and skips the walk over the map: run(prepared, eps) is the bare bn_bp_run_view call.
    struct evidence_arrays {
        std::vector<std::int32_t> node, off;
        std::vector<double> val;
    };
    evidence_arrays prepare(std::unordered_map<vertex_type, matrix_type> const& precondition)
    {
        evidence_arrays out;
        marshal(precondition, out);
        return out;
    }
    view_type run(evidence_arrays const& evidence, double const epsilon = 0.001)
    {
        // the marginals arrive in a page-locked buffer the engine owns (one DMA behind the run, one
        // synchronisation for upload + run + download)
        double const* beliefs = nullptr;
        std::int32_t sweeps = 0;
        double residual = 0;
        static std::int32_t const zero = 0;
        mi355x::engine_handle::check(bn_bp_run_view(
            engine_.get(), static_cast<std::int32_t>(evidence.node.size()), evidence.node.data(),
            evidence.off.empty() ? &zero : evidence.off.data(), evidence.val.data(),
            epsilon, 0 /* unbounded, like the reference */, &beliefs, &sweeps, &residual));
        last_sweeps_ = sweeps;
        last_residual_ = residual;
        return view_type(model_, beliefs);
    }

    // Not in the reference (one query per operator() call, :31): several evidence sets on this network in ONE
    // call.  Entry q of the result is exactly what operator()(queries[q], epsilon) returns -- same iteration
    // count, same bits -- but the queries share every kernel launch (bn_bp_run_batch, bn_mi355x.h).
    // Up to BN_MAX_BATCH_SETS queries per call; longer lists are processed in slices of that size.
    std::vector<return_type> run_batch(std::vector<std::unordered_map<vertex_type, matrix_type>> const& queries,
                                       double const epsilon = 0.001)
    {
        std::vector<return_type> results;
        results.reserve(queries.size());
        std::size_t const nbel = static_cast<std::size_t>(model_.node_off.back());
        for(std::size_t begin = 0; begin < queries.size(); begin += BN_MAX_BATCH_SETS)
        {
            std::size_t const count = std::min<std::size_t>(BN_MAX_BATCH_SETS, queries.size() - begin);
            std::vector<std::int32_t> ne, ev_node, ev_off;
            std::vector<double> ev_val;
            for(std::size_t q = 0; q < count; ++q)
            {
                std::size_t const val_begin = ev_val.size();
                ev_off.push_back(0);  // every set's offsets start at 0
                for(auto const& p : queries[begin + q])
                {
                    auto const it = model_.index.find(p.first);
                    if(it == model_.index.end()) throw std::runtime_error("belief_propagation: evidence on an unknown vertex");
                    if(p.second.height() != 1) throw std::runtime_error("belief_propagation: evidence must be a 1 x k matrix");
                    ev_node.push_back(it->second);
                    ev_val.insert(ev_val.end(), p.second[0].begin(), p.second[0].end());
                    ev_off.push_back(static_cast<std::int32_t>(ev_val.size() - val_begin));
                }
                ne.push_back(static_cast<std::int32_t>(queries[begin + q].size()));
            }
            std::vector<double> beliefs(nbel * count);
            std::vector<std::int32_t> sweeps(count, 0);
            std::vector<double> residual(count, 0.0);
            mi355x::engine_handle::check(bn_bp_run_batch(
                engine_.get(), static_cast<std::int32_t>(count), ne.data(), ev_node.data(), ev_off.data(), ev_val.data(),
                epsilon, 0, beliefs.data(), sweeps.data(), residual.data()));
            for(std::size_t q = 0; q < count; ++q)
            {
                results.push_back(view_type(model_, beliefs.data() + q * nbel).to_map());
            }
            last_sweeps_ = sweeps.back();
            last_residual_ = residual.back();
        }
        return results;
    }

    // DIFFERENCE FROM THE REFERENCE: it reads node->cpt at every call (:61, :186, :252), so a table edited or re-fitted
    // (sampler::make_cpt) after the functor was built is seen by the next operator().  This functor flattens the tables ONCE, in
    // its constructor, into device images.  reload() brings them up to date -- through the functor's own copy of the graph
    // (vertices are shared_ptrs: it sees the caller's edits, like the reference's graph_ member), or through a graph passed
    // in; the structure (vertices, arities, edges) must be unchanged.  One pass over the tables + one host-to-device copy.
    void reload() { mi355x::reload_cpts(graph_, model_, engine_); }
    void reload(graph_t const& graph) { mi355x::reload_cpts(graph, model_, engine_); graph_ = graph; }

    // Not in the reference (its operator() hides them): iterations and last maximum_difference.
    int last_sweeps() const { return last_sweeps_; }
    double last_residual() const { return last_residual_; }

private:
    // precondition -> evidence_arrays
    void marshal(std::unordered_map<vertex_type, matrix_type> const& precondition, evidence_arrays& out)
    {
        // Scratch kept between calls: a query allocates nothing once the vectors have grown.  The caller's map is a linked
        // list of heap nodes, each pointing at a row table that points at a row: walked entry by entry that is four
        // dependent cache misses per evidence node (0.12 ms for 998 of them on the 99 856-node grid, half of the GPU's share).
        // So it is walked in STAGES -- the list itself, then all row tables, then all rows -- each a loop of independent
        // loads the core overlaps, with the position-table line of every vertex requested in the first.
        out.node.clear();
        out.val.clear();
        out.off.assign(1, 0);
        ev_key_.clear();
        ev_mat_.clear();
        ev_row_.clear();
        for(auto const& p : precondition)
        {
            model_.lookup.prefetch(p.first.get());
            ev_key_.push_back(p.first.get());
            ev_mat_.push_back(&p.second);
        }
        for(matrix_type const* m : ev_mat_)
        {
            if(m->height() != 1) throw std::runtime_error("belief_propagation: evidence must be a 1 x k matrix");
            ev_row_.push_back(&(*m)[0]);
        }
        for(std::size_t j = 0; j < ev_key_.size(); ++j)
        {
            std::int32_t const position = model_.lookup.find(ev_key_[j]);
            if(position < 0) throw std::runtime_error("belief_propagation: evidence on an unknown vertex");
            out.node.push_back(position);
            out.off.push_back(out.off.back() + static_cast<std::int32_t>(ev_row_[j]->size()));
        }
        out.val.resize(static_cast<std::size_t>(out.off.back()));
        for(std::size_t j = 0; j < ev_row_.size(); ++j)
            std::copy(ev_row_[j]->begin(), ev_row_[j]->end(), out.val.begin() + out.off[j]);
    }

    graph_t graph_;   // the reference keeps a copy too (graph_t const graph_, :320)
    mi355x::flat_model model_;
    mi355x::engine_handle engine_;
    evidence_arrays scratch_;   // evidence marshalling scratch of run(precondition)
    std::vector<void const*> ev_key_;
    std::vector<matrix_type const*> ev_mat_;
    std::vector<std::vector<double> const*> ev_row_;
    int last_sweeps_ = 0;
    double last_residual_ = 0;
};

} // namespace inference
} // namespace bn

#endif // #ifndef BNI_INFERENCE_BELIEF_PROPAGATION_HPP

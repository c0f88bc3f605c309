// bayesian/inference/rejection_sampling.hpp -- MI355X drop-in for the reference header of the same
// path (reference rejection_sampling.hpp:13-62):
//
//     bn::inference::rejection_sampling rs(graph);
//     auto marginals = rs({{vertex_4, 1}, {vertex_1, 0}}, 10000);   // logic sampling
//
// Forward samples are drawn on the GPU (bn_rs_run) until generate_sample_num of them agree with
// the condition; marginals are the plain state frequencies over those accepted samples.  The
// reference never returns when the condition has probability zero; this functor gives up after
// max_draws() samples and throws std::runtime_error.
#ifndef BNI_INFERENCE_REJECTION_SAMPLING_HPP
#define BNI_INFERENCE_REJECTION_SAMPLING_HPP

#include <cstdint>
#include <random>
#include <unordered_map>
#include <utility>
#include <vector>

#include "mi355x_flatten.hpp"
#include "mi355x_marginals.hpp"

namespace bn {
namespace inference {

class rejection_sampling {
public:
    typedef std::unordered_map<vertex_type, matrix_type> return_type;
    typedef std::vector<std::unordered_map<vertex_type, int>> pattern_list;

    explicit rejection_sampling(graph_t const& graph)
        : graph_(graph), model_(mi355x::flatten(graph)), engine_(model_)
    {
        std::random_device rand_dev;
        seed_ = (static_cast<std::uint64_t>(rand_dev()) << 32) ^ rand_dev();
    }

    virtual ~rejection_sampling() = default;

    // DIFFERENCE FROM THE REFERENCE: it reads node->cpt while it samples, so tables edited or re-fitted after the functor was built
    // are seen by the next call; this functor flattened them once, in its constructor.  reload() brings the device images up to
    // date (same structure required) -- through the functor's own copy of the graph, whose vertices are the caller's.
    void reload() { mi355x::reload_cpts(graph_, model_, engine_); }
    void reload(graph_t const& graph) { mi355x::reload_cpts(graph, model_, engine_); graph_ = graph; }

    void seed(std::uint64_t s) { seed_ = s; next_sample_ = 0; }
    void max_draws(std::uint64_t m) { max_draws_ = m; }
    std::uint64_t last_drawn() const { return last_drawn_; }

    // By-pass (reference :26-30)
    inline return_type operator()(int const generate_sample_num = 10000)
    {
        std::vector<std::pair<vertex_type, int>> const condition;
        return operator()(condition, generate_sample_num);
    }

    // Run: Logic Sampling (a.k.a. Rejection Sampling) (reference :33-62)
    return_type operator()(std::vector<std::pair<vertex_type, int>> const& condition, int const generate_sample_num = 10000)
    {
        return run(condition, generate_sample_num).to_map();
    }

    // Not in the reference: the same call with the marginals read in place (mi355x_marginals.hpp) instead of the
    // reference's map of 1 x k matrices; valid until the next call on this functor.
    typedef mi355x::marginals_view view_type;
    view_type run(std::vector<std::pair<vertex_type, int>> const& condition, int const generate_sample_num = 10000)
    {
        std::vector<std::int32_t> ev_node, ev_state;
        for(auto const& c : condition)
        {
            auto const it = model_.index.find(c.first);
            // the reference finds no such key in any pattern and rejects every sample (:76-77)
            if(it == model_.index.end()) throw std::runtime_error("rejection_sampling: condition on an unknown vertex");
            ev_node.push_back(it->second);
            ev_state.push_back(c.second);
        }
        marginals_.resize(static_cast<std::size_t>(model_.node_off.back()));
        std::uint64_t drawn = 0, accepted = 0;
        mi355x::engine_handle::check(bn_rs_run(
            engine_.get(), static_cast<std::int32_t>(ev_node.size()), ev_node.data(), ev_state.data(), next_sample_,
            static_cast<std::uint64_t>(generate_sample_num), max_draws_, seed_, marginals_.data(), &drawn, &accepted));
        next_sample_ += drawn;
        last_drawn_ = drawn;
        if(accepted < static_cast<std::uint64_t>(generate_sample_num))
            throw std::runtime_error("rejection_sampling: condition too unlikely, gave up after max_draws() samples");
        for(double& c : marginals_) c = c / static_cast<double>(accepted);
        return view_type(model_, marginals_.data());
    }

private:
    graph_t graph_;   // the reference keeps a copy too
    mi355x::flat_model model_;
    mi355x::engine_handle engine_;
    std::vector<double> marginals_;   // what run()'s view reads
    std::uint64_t seed_ = 0, next_sample_ = 0, last_drawn_ = 0;
    std::uint64_t max_draws_ = std::uint64_t(1) << 34;
};

} // namespace inference
} // namespace bn

#endif // #ifndef BNI_INFERENCE_REJECTION_SAMPLING_HPP

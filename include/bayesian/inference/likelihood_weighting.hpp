// bayesian/inference/likelihood_weighting.hpp -- MI355X drop-in for the reference header of the
// same path (reference likelihood_weighting.hpp:13-59):
//
//     bn::inference::likelihood_weighting lw(graph);
//     auto marginals = lw(evidence, sample_num);        // sample_num defaults to 10000
//
// Sampling runs in the HIP kernel behind bn_lw_run.  The reference seeds an mt19937 from
// std::random_device (:224-244), so its stream is not reproducible by design; this functor draws
// its seed the same way once, then numbers its samples consecutively (each sample id owns a Philox-seeded xoshiro128++ stream) --
// successive calls continue the numbering.
#ifndef BNI_INFERENCE_LIKELIHOOD_WEIGHTING_HPP
#define BNI_INFERENCE_LIKELIHOOD_WEIGHTING_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <random>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "mi355x_flatten.hpp"
#include "mi355x_marginals.hpp"

namespace bn {
namespace inference {

class likelihood_weighting {
public:
    typedef std::unordered_map<vertex_type, int> evidence_list;
    typedef std::unordered_map<vertex_type, int> pattern_list;
    typedef std::unordered_map<bn::condition_t, std::size_t> sample_list;
    typedef std::unordered_map<vertex_type, matrix_type> return_type;

    explicit likelihood_weighting(graph_t const& graph)
        : graph_(graph), model_(mi355x::flatten(graph)), engine_(model_)
    {
        std::random_device rand_dev;
        seed_ = (static_cast<std::uint64_t>(rand_dev()) << 32) ^ rand_dev();
    }

    virtual ~likelihood_weighting() = default;

    // DIFFERENCE FROM THE REFERENCE: it reads node->cpt while it samples, so tables edited or re-fitted after the functor was built
    // are seen by the next call; this functor flattened them once, in its constructor.  reload() brings the device images up to
    // date (same structure required) -- through the functor's own copy of the graph, whose vertices are the caller's.
    void reload() { mi355x::reload_cpts(graph_, model_, engine_); }
    void reload(graph_t const& graph) { mi355x::reload_cpts(graph, model_, engine_); graph_ = graph; }

    // deterministic runs (tests)
    void seed(std::uint64_t s) { seed_ = s; next_sample_ = 0; }
    // units the last make_samples call executed (the reference does not report it; tests compare it
    // with the oracle's count)
    std::uint64_t last_units() const { return last_units_; }

    // Run: Likelihood Weighting (reference :28-59)
    return_type operator()(evidence_list const& evidence, std::uint64_t const sample_num = 10000)
    {
        return run(evidence, sample_num).to_map();
    }

    // Not in the reference: the same call with the marginals read in place (mi355x_marginals.hpp) instead of the
    // reference's map of 1 x k matrices (three heap blocks per node); valid until the next call on this functor.
    typedef mi355x::marginals_view view_type;
    view_type run(evidence_list const& evidence, std::uint64_t const sample_num = 10000)
    {
        std::vector<std::int32_t> ev_node, ev_state;
        for(auto const& e : evidence)
        {
            auto const it = model_.index.find(e.first);
            if(it == model_.index.end()) throw std::runtime_error("likelihood_weighting: evidence on an unknown vertex");
            ev_node.push_back(it->second);
            ev_state.push_back(e.second);
        }
        marginals_.resize(static_cast<std::size_t>(model_.node_off.back()));
        mi355x::engine_handle::check(bn_lw_run(
            engine_.get(), static_cast<std::int32_t>(ev_node.size()), ev_node.data(), ev_state.data(), next_sample_,
            sample_num, seed_, marginals_.data()));
        next_sample_ += sample_num;

        for(std::size_t i = 0; i < model_.nodes.size(); ++i)
        {
            std::size_t const kv = static_cast<std::size_t>(model_.k[i]);
            double* const h = marginals_.data() + model_.node_off[i];
            // Normalization, reference :197-221: uniform when the weights vanish
            double sum = 0;
            for(std::size_t j = 0; j < kv; ++j) sum += h[j];
            for(std::size_t j = 0; j < kv; ++j) h[j] = (sum < 1.0e-20) ? 1.00 / kv : h[j] / sum;
        }
        return view_type(model_, marginals_.data());
    }

    // Make: accurate sample (reference :62-117).  Units of unit_size weighted samples are drawn
    // until no normalised marginal moved by epsilon or more between two consecutive units; returns
    // the (unweighted) histogram of complete joint patterns over all units and the marginals.
    // The patterns are read back from the GPU's state matrix, so unit_size x nodes must fit the
    // download budget (4 GiB) -- the joint histogram is meaningless for large networks anyway.
    std::pair<sample_list, return_type> make_samples(
        evidence_list const& evidence,
        std::uint64_t const unit_size = 1000000/* 1'000'000 */,
        double const epsilon = 0.001
        )
    {
        std::size_t const n = model_.nodes.size();
        if(static_cast<double>(unit_size) * static_cast<double>(n) > 4294967296.0)
            throw std::runtime_error("likelihood_weighting::make_samples: unit_size x nodes exceeds 4 GiB of states");
        std::vector<std::int32_t> ev_node, ev_state;
        for(auto const& e : evidence)
        {
            auto const it = model_.index.find(e.first);
            if(it == model_.index.end()) throw std::runtime_error("likelihood_weighting: evidence on an unknown vertex");
            ev_node.push_back(it->second);
            ev_state.push_back(e.second);
        }
        std::size_t const hn = static_cast<std::size_t>(model_.node_off.back());
        std::vector<double> w_list(hn, 0.0), unit_hist(hn), probabilities(hn, 0.0), next(hn);
        std::unordered_map<std::string, std::size_t> packed;   // pattern bytes -> occurrences
        std::vector<std::uint8_t> states(static_cast<std::size_t>(std::min<std::uint64_t>(unit_size, std::uint64_t(1) << 20)) * n);

        last_units_ = 0;
        while(true)
        {
            // Generate one unit (:85-99), in pieces the device keeps in one state matrix
            for(std::uint64_t done = 0; done < unit_size;)
            {
                std::uint64_t const piece = std::min<std::uint64_t>(unit_size - done, std::uint64_t(1) << 20);
                mi355x::engine_handle::check(bn_lw_run(
                    engine_.get(), static_cast<std::int32_t>(ev_node.size()), ev_node.data(), ev_state.data(),
                    next_sample_, piece, seed_, unit_hist.data()));
                next_sample_ += piece;
                for(std::size_t i = 0; i < hn; ++i) w_list[i] += unit_hist[i];
                mi355x::engine_handle::check(bn_lw_states(engine_.get(), piece, states.data(), nullptr));
                for(std::uint64_t s = 0; s < piece; ++s)
                    ++packed[std::string(reinterpret_cast<char const*>(states.data() + s * n), n)];
                done += piece;
            }

            ++last_units_;
            // largest move of any normalised marginal since the previous unit (:101-112)
            double max_difference = std::numeric_limits<double>::min();
            for(std::size_t v = 0; v < n; ++v)
            {
                std::size_t const kv = static_cast<std::size_t>(model_.k[v]);
                double sum = 0;
                for(std::size_t j = 0; j < kv; ++j) sum += w_list[model_.node_off[v] + j];
                for(std::size_t j = 0; j < kv; ++j)
                {
                    std::size_t const at = static_cast<std::size_t>(model_.node_off[v]) + j;
                    next[at] = (sum < 1.0e-20) ? 1.00 / kv : w_list[at] / sum;   // normalize, :197-221
                    max_difference = std::max(max_difference, std::abs(probabilities[at] - next[at]));
                }
            }
            probabilities = next;
            if(max_difference < epsilon) break;
        }

        sample_list patterns;
        for(auto const& kv : packed)
        {
            bn::condition_t pattern;
            for(std::size_t v = 0; v < n; ++v) pattern[model_.nodes[v]] = static_cast<unsigned char>(kv.first[v]);
            patterns[pattern] = kv.second;
        }
        return std::make_pair(std::move(patterns), view_type(model_, probabilities.data()).to_map());
    }

private:
    graph_t graph_;   // the reference keeps a copy too
    mi355x::flat_model model_;
    mi355x::engine_handle engine_;
    std::vector<double> marginals_;   // what run()'s view reads
    std::uint64_t seed_ = 0;
    std::uint64_t next_sample_ = 0;
    std::uint64_t last_units_ = 0;
};

} // namespace inference
} // namespace bn

#endif // #ifndef BNI_INFERENCE_LIKELIHOOD_WEIGHTING_HPP

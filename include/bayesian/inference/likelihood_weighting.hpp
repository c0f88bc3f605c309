// bayesian/inference/likelihood_weighting.hpp -- MI355X drop-in for the reference header of the
// same path (reference likelihood_weighting.hpp:13-59):
//
//     bn::inference::likelihood_weighting lw(graph);
//     auto marginals = lw(evidence, sample_num);        // sample_num defaults to 10000
//
// Sampling runs in the HIP kernel behind bn_lw_run.  The reference seeds an mt19937 from
// std::random_device (:224-244), so its stream is not reproducible by design; this functor draws
// its seed the same way once, then walks a Philox4x32-10 stream -- successive calls continue it.
#ifndef BNI_INFERENCE_LIKELIHOOD_WEIGHTING_HPP
#define BNI_INFERENCE_LIKELIHOOD_WEIGHTING_HPP

#include <cstdint>
#include <random>
#include <unordered_map>
#include <vector>

#include "mi355x_flatten.hpp"

namespace bn {
namespace inference {

class likelihood_weighting {
public:
    typedef std::unordered_map<vertex_type, int> evidence_list;
    typedef std::unordered_map<vertex_type, int> pattern_list;
    typedef std::unordered_map<bn::condition_t, std::size_t> sample_list;
    typedef std::unordered_map<vertex_type, matrix_type> return_type;

    explicit likelihood_weighting(graph_t const& graph)
        : model_(mi355x::flatten(graph)), engine_(model_)
    {
        std::random_device rand_dev;
        seed_ = (static_cast<std::uint64_t>(rand_dev()) << 32) ^ rand_dev();
    }

    virtual ~likelihood_weighting() = default;

    // deterministic runs (tests)
    void seed(std::uint64_t s) { seed_ = s; next_sample_ = 0; }

    // Run: Likelihood Weighting (reference :28-59)
    return_type operator()(evidence_list const& evidence, std::uint64_t const sample_num = 10000)
    {
        std::vector<std::int32_t> ev_node, ev_state;
        for(auto const& e : evidence)
        {
            auto const it = model_.index.find(e.first);
            if(it == model_.index.end()) throw std::runtime_error("likelihood_weighting: evidence on an unknown vertex");
            ev_node.push_back(it->second);
            ev_state.push_back(e.second);
        }
        std::vector<double> hist(static_cast<std::size_t>(model_.node_off.back()));
        mi355x::engine_handle::check(bn_lw_run(
            engine_.get(), static_cast<std::int32_t>(ev_node.size()), ev_node.data(), ev_state.data(), next_sample_,
            sample_num, seed_, hist.data()));
        next_sample_ += sample_num;

        return_type ret;
        for(std::size_t i = 0; i < model_.nodes.size(); ++i)
        {
            std::size_t const kv = static_cast<std::size_t>(model_.k[i]);
            matrix_type m(1, kv);
            // Normalization, reference :197-221: uniform when the weights vanish
            double sum = 0;
            for(std::size_t j = 0; j < kv; ++j) sum += hist[model_.node_off[i] + j];
            for(std::size_t j = 0; j < kv; ++j)
                m[0][j] = (sum < 1.0e-20) ? 1.00 / kv : hist[model_.node_off[i] + j] / sum;
            ret[model_.nodes[i]] = m;
        }
        return ret;
    }

private:
    mi355x::flat_model model_;
    mi355x::engine_handle engine_;
    std::uint64_t seed_ = 0;
    std::uint64_t next_sample_ = 0;
};

} // namespace inference
} // namespace bn

#endif // #ifndef BNI_INFERENCE_LIKELIHOOD_WEIGHTING_HPP

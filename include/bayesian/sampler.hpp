// bayesian/sampler.hpp -- drop-in for the reference's bn::sampler (bayesian/sampler.hpp:17-215)
// whose make_cpt() runs on the MI355X through bn_fit_cpt (include/bn_mi355x.h).
//
// Same class, same members: sampler(), sampler(filename), load_sample(table),
// load_sample(node_list), make_cpt(graph), filename(), set_filename(), table(), sampling_size().
// load_sample(node_list) reads "count s_0 s_1 ..." rows split on runs of whitespace -- what the
// reference does with boost::algorithm::split + token_compress_on (:57-70) -- without Boost.
// make_cpt (:81-163): cpt.assign() on every node, then per node and parent assignment the counts
// of the patterns summed by own state and normalised; a row no pattern supports is uniform.  A
// pattern lacking one of the graph's nodes throws std::out_of_range like condition_t::at (:108,:119).
#ifndef BNI_SAMPLER_HPP
#define BNI_SAMPLER_HPP

#include <cstdint>
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include <bayesian/graph.hpp>
#include <bayesian/inference/mi355x_flatten.hpp>

namespace bn {

class sampler {
public:
    sampler() : filename_(), table_(), sampling_size_(0) {}
    sampler(std::string const& filename) : filename_(filename), table_(), sampling_size_(0) {}

    bool load_sample(std::unordered_map<condition_t, std::size_t> const& table)
    {
        table_ = table;
        sampling_size_ = 0;
        for(auto const& p : table) sampling_size_ += p.second;
        return true;
    }

    // Sample file (reference sampler.hpp:42-77): one row per distinct pattern, whitespace separated,
    //   <occurrences> <state of node_list[0]> <state of node_list[1]> ...
    // Rows with the same pattern add up.  False when the file cannot be opened; a row with fewer
    // fields than nodes throws std::out_of_range.
    bool load_sample(std::vector<vertex_type> const& node_list)
    {
        std::ifstream file(filename_);
        if(!file.is_open()) return false;

        std::unordered_map<condition_t, std::size_t> counted;
        std::size_t total = 0;
        std::size_t const fields = node_list.size() + 1;
        for(std::string row; std::getline(file, row);)
        {
            std::istringstream in(row);
            std::vector<long> value;
            for(long x; in >> x;) value.push_back(x);
            if(value.size() < fields) throw std::out_of_range("bn::sampler: short sample row");

            condition_t pattern;
            for(std::size_t col = 1; col < fields; ++col) pattern[node_list[col - 1]] = static_cast<int>(value[col]);
            std::size_t const occurrences = static_cast<std::size_t>(value[0]);
            counted[pattern] += occurrences;
            total += occurrences;
        }
        table_.swap(counted);
        sampling_size_ = total;
        return true;
    }

    bool make_cpt(graph_t const& graph) const
    {
        if(sampling_size() == 0) return false;

        auto const nodes = graph.vertex_list();
        for(auto const& node : nodes) node->cpt.assign(graph.in_vertexes(node), node);

        mi355x::flat_model const fm = mi355x::flatten_structure(graph);
        std::size_t const n = nodes.size();
        std::vector<std::uint8_t> patterns;
        std::vector<std::uint64_t> counts;
        patterns.reserve(table_.size() * n);
        counts.reserve(table_.size());
        for(auto const& sample : table_)
        {
            for(auto const& node : nodes) patterns.push_back(static_cast<std::uint8_t>(sample.first.at(node)));
            counts.push_back(sample.second);
        }
        std::vector<double> cpt(static_cast<std::size_t>(fm.cpt_off.back()));
        bn_model_desc const d = fm.desc();
        mi355x::engine_handle::check(bn_fit_cpt(
            &d, static_cast<std::int64_t>(counts.size()), patterns.data(), counts.data(), cpt.data()));
        mi355x::store_cpts(graph, fm, cpt);
        return true;
    }

    std::string filename() const { return filename_; }

    void set_filename(std::string const& filename)
    {
        filename_ = filename;
        sampling_size_ = 0;
        table_.clear();
    }

    std::unordered_map<condition_t, std::size_t> table() const { return table_; }

    std::size_t sampling_size() const { return sampling_size_; }

private:
    std::string filename_;
    std::unordered_map<condition_t, std::size_t> table_;
    std::size_t sampling_size_;
};

} // namespace bn

#endif // BNI_SAMPLER_HPP

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

    bool load_sample(std::vector<vertex_type> const& node_list)
    {
        std::ifstream ifs(filename_);
        if(!ifs.is_open()) return false;
        std::size_t sampling_size = 0;
        std::unordered_map<condition_t, std::size_t> table;
        std::string line_str;
        condition_t sample;
        while(std::getline(ifs, line_str))
        {
            std::istringstream iss(line_str);
            std::vector<std::string> line;
            for(std::string tok; iss >> tok;) line.push_back(tok);
            if(line.size() < node_list.size() + 1) throw std::out_of_range("bn::sampler: short sample row");
            for(std::size_t i = 0; i < node_list.size(); ++i) sample[node_list[i]] = std::stoi(line[i + 1]);
            auto const sample_num = static_cast<std::size_t>(std::stoi(line[0]));
            table[sample] += sample_num;
            sampling_size += sample_num;
        }
        sampling_size_ = sampling_size;
        table_ = std::move(table);
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

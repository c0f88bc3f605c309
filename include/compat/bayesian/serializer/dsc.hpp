// compat/bayesian/serializer/dsc.hpp -- this repository's own reader for the DSC dialect that the
// reference's loader accepts (bayesian/serializer/dsc.hpp:33-227): same class name and entry
// points (parse / from_file / from_data), independent implementation (tokenises each block instead
// of slicing fixed columns).  For builds where the reference is not installed; with the reference
// on the include path its own loader is used and works unchanged with the drop-in inference headers.
#ifndef BNI_SERIALIZER_DSC_HPP
#define BNI_SERIALIZER_DSC_HPP

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include <bayesian/graph.hpp>

namespace bn {
namespace serializer {

class dsc {
public:
    graph_t parse(std::vector<std::string> data)
    {
        std::vector<std::string> lines;
        for(auto& raw : data)
        {
            std::string const t = trim(raw);
            if(!t.empty() && t.compare(0, 2, "//") != 0) lines.push_back(t);
        }
        graph_t graph;
        std::unordered_map<std::string, vertex_type> by_name;
        std::size_t i = 0;
        while(i < lines.size())
        {
            std::string const& ln = lines[i];
            if(ln.compare(0, 5, "node ") == 0)
            {
                std::string const name = trim(ln.substr(5));
                auto v = graph.add_vertex();
                by_name[name] = v;
                i = expect_open(lines, i + 1);
                for(; i < lines.size() && lines[i] != "}"; ++i)
                {
                    auto const lb = lines[i].find('['), rb = lines[i].find(']');
                    if(lines[i].compare(0, 4, "type") == 0 && lb != std::string::npos && rb != std::string::npos)
                        v->selectable_num = static_cast<std::size_t>(std::atoi(lines[i].substr(lb + 1, rb - lb - 1).c_str()));
                }
                ++i;
            }
            else if(ln.compare(0, 11, "probability") == 0)
            {
                auto const lp = ln.find('('), rp = ln.rfind(')');
                if(lp == std::string::npos || rp == std::string::npos) throw std::runtime_error("dsc: bad probability line");
                std::string const inside = ln.substr(lp + 1, rp - lp - 1);
                auto const bar = inside.find('|');
                std::string const target_name = trim(inside.substr(0, bar));
                std::vector<vertex_type> parents;
                if(bar != std::string::npos)
                    for(auto const& nm : split(inside.substr(bar + 1), ','))
                        parents.push_back(lookup(by_name, trim(nm)));
                vertex_type const target = lookup(by_name, target_name);
                for(auto const& p : parents) graph.add_edge(p, target);
                target->cpt.assign(parents, target);
                i = expect_open(lines, i + 1);
                for(; i < lines.size() && lines[i] != "}"; ++i)
                {
                    std::string row = lines[i];
                    if(!row.empty() && row.back() == ';') row.pop_back();
                    condition_t cond;
                    std::string values = row;
                    if(!parents.empty())
                    {
                        auto const a = row.find('('), b = row.find(')'), c = row.find(':');
                        if(a == std::string::npos || b == std::string::npos || c == std::string::npos)
                            throw std::runtime_error("dsc: bad CPT row");
                        auto const states = split(row.substr(a + 1, b - a - 1), ',');
                        if(states.size() != parents.size()) throw std::runtime_error("dsc: row key size");
                        for(std::size_t j = 0; j < parents.size(); ++j) cond[parents[j]] = std::atoi(states[j].c_str());
                        values = row.substr(c + 1);
                    }
                    std::vector<double> probs;
                    for(auto const& tok : split(values, ',')) probs.push_back(std::strtod(tok.c_str(), nullptr));
                    auto entry = target->cpt[cond];
                    if(!entry.first) throw std::runtime_error("dsc: CPT row names an impossible assignment");
                    entry.second = probs;
                }
                ++i;
            }
            else
            {
                ++i;  // header line and anything unknown
            }
        }
        return graph;
    }

    graph_t from_file(std::string const& filename)
    {
        std::ifstream ifs(filename);
        if(!ifs) throw std::runtime_error("dsc: cannot open " + filename);
        return parse(read_lines(ifs));
    }

    graph_t from_data(std::string const& data)
    {
        std::istringstream iss(data);
        return parse(read_lines(iss));
    }

private:
    template<class Stream> static std::vector<std::string> read_lines(Stream& is)
    {
        std::vector<std::string> out;
        for(std::string ln; std::getline(is, ln);) out.push_back(ln);
        return out;
    }
    static std::string trim(std::string const& s)
    {
        std::size_t a = 0, b = s.size();
        while(a < b && std::isspace(static_cast<unsigned char>(s[a]))) ++a;
        while(b > a && std::isspace(static_cast<unsigned char>(s[b - 1]))) --b;
        return s.substr(a, b - a);
    }
    static std::vector<std::string> split(std::string const& s, char sep)
    {
        std::vector<std::string> out;
        std::string cur;
        for(char c : s)
        {
            if(c == sep) { out.push_back(trim(cur)); cur.clear(); }
            else cur.push_back(c);
        }
        out.push_back(trim(cur));
        return out;
    }
    static std::size_t expect_open(std::vector<std::string> const& lines, std::size_t i)
    {
        if(i >= lines.size() || lines[i] != "{") throw std::runtime_error("dsc: '{' expected on its own line");
        return i + 1;
    }
    static vertex_type lookup(std::unordered_map<std::string, vertex_type> const& m, std::string const& name)
    {
        auto const it = m.find(name);
        if(it == m.end()) throw std::runtime_error("dsc: unknown node " + name);
        return it->second;
    }
};

} // namespace serializer
} // namespace bn

#endif // BNI_SERIALIZER_DSC_HPP

// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Runs the UNMODIFIED reference headers (included from /root/reference where they
// lie; nothing is copied into this repository) on a flat model read from a text
// file, and prints the reference's outputs as JSON.  Built by oracle/Makefile into
// oracle/_ref/ref_driver (git-ignored).  Used to
//   * generate tests/golden/*.json (tests/golden/make_golden.py), and
//   * validate oracle/bp_oracle.c and oracle/lw_oracle.c in this container.
//
// `private` is made public so that the BP sweep can be stepped one iteration at a
// time (to record the sweep count and the per-sweep residual, which
// belief_propagation::operator() does not expose) and so that the likelihood
// weighting engine can be reseeded deterministically.  The stepped result is
// checked against a plain operator() call on a second instance: "stepped_equals_call".
//
// Input (whitespace separated):
//   BNFLAT1  n  k[0..n)  { m p[0..m) } x n   { cpt row-major } x n
//   then either   bp  eps  dump_msgs  ne  { node kv val[0..kv) } x ne
//   or            lw  n_samples seed  ne  { node state } x ne
//   or            lwms unit_size eps seed  ne  { node state } x ne     (likelihood_weighting::make_samples)
//   or            rs  n_accept seed  ne  { node state } x ne           (rejection_sampling::operator())
// DSC mode:  ref_driver --dsc net.dsc request.txt   -- the network comes from the reference's
//   serializer::dsc loader (dsc.hpp:71), request.txt holds the "bp ..." / "lw ..." part; the JSON
//   additionally carries the flat model as this repository's flatten() sees the loaded graph.
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iostream>
#include <limits>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#define private public
#include <bayesian/graph.hpp>
#include <bayesian/matrix.hpp>
#include <bayesian/inference/belief_propagation.hpp>
#include <bayesian/inference/likelihood_weighting.hpp>
#include <bayesian/inference/rejection_sampling.hpp>
#undef private
#include <bayesian/serializer/dsc.hpp>                               // the reference's own loader
#include "../include/bayesian/inference/mi355x_flatten.hpp"          // this repo: graph_t -> flat arrays

struct flat_model {
    int n = 0;
    std::vector<int> k;
    std::vector<std::vector<int>> parents;
    std::vector<std::vector<double>> cpt;
};

static void die(const char* msg) { std::fprintf(stderr, "ref_driver: %s\n", msg); std::exit(2); }

static bn::graph_t build_graph(flat_model const& fm)
{
    bn::graph_t g;
    for(int v = 0; v < fm.n; ++v) {
        auto vx = g.add_vertex();
        vx->id = v;
        vx->selectable_num = fm.k[v];
    }
    auto const& vl = g.vertex_list();
    // all in-edges of v before v gets any child (parents have lower ids in our models),
    // so graph_t::is_able_trace stays cheap
    for(int v = 0; v < fm.n; ++v)
        for(int p : fm.parents[v])
            if(!g.add_edge(vl[p], vl[v])) die("add_edge refused (cycle or duplicate)");
    for(int v = 0; v < fm.n; ++v) {
        std::vector<bn::vertex_type> ps;
        for(int p : fm.parents[v]) ps.push_back(vl[p]);
        vl[v]->cpt.assign(ps, vl[v]);
        std::size_t rows = 1;
        for(int p : fm.parents[v]) rows *= fm.k[p];
        std::vector<int> st(ps.size(), 0);
        for(std::size_t r = 0; r < rows; ++r) {
            bn::condition_t cond;
            for(std::size_t j = 0; j < ps.size(); ++j) cond[ps[j]] = st[j];
            std::vector<double> row(fm.cpt[v].begin() + r * fm.k[v], fm.cpt[v].begin() + (r + 1) * fm.k[v]);
            auto res = vl[v]->cpt[cond];
            if(!res.first) die("cpt row missing after assign");
            res.second = row;
            for(int j = (int)ps.size() - 1; j >= 0; --j) { // first parent most significant
                if(++st[j] < fm.k[fm.parents[v][j]]) break;
                st[j] = 0;
            }
        }
    }
    return g;
}

static void print_vec(std::vector<double> const& v)
{
    std::printf("[");
    for(std::size_t i = 0; i < v.size(); ++i) std::printf("%s%.17g", i ? "," : "", v[i]);
    std::printf("]");
}

int main(int argc, char** argv)
{
    if(argc < 2) die("usage: ref_driver model.txt | --dsc net.dsc request.txt");
    bool const dsc_mode = std::string(argv[1]) == "--dsc";
    if(dsc_mode && argc < 4) die("usage: ref_driver --dsc net.dsc request.txt");
    std::ifstream in(dsc_mode ? argv[3] : argv[1]);
    if(!in) die("cannot open input");
    flat_model fm;
    bn::graph_t graph;
    auto t0 = std::chrono::steady_clock::now();
    if(dsc_mode)
    {
        graph = bn::serializer::dsc().from_file(argv[2]);
        auto const flat = bn::mi355x::flatten(graph);
        fm.n = static_cast<int>(flat.k.size());
        fm.k.assign(flat.k.begin(), flat.k.end());
        fm.parents.resize(fm.n); fm.cpt.resize(fm.n);
        for(int v = 0; v < fm.n; ++v) {
            fm.parents[v].assign(flat.in_idx.begin() + flat.in_ptr[v], flat.in_idx.begin() + flat.in_ptr[v + 1]);
            fm.cpt[v].assign(flat.cpt.begin() + flat.cpt_off[v], flat.cpt.begin() + flat.cpt_off[v + 1]);
        }
    }
    else
    {
        std::string magic;
        in >> magic;
        if(magic != "BNFLAT1") die("bad magic");
        in >> fm.n;
        fm.k.resize(fm.n); fm.parents.resize(fm.n); fm.cpt.resize(fm.n);
        for(int v = 0; v < fm.n; ++v) in >> fm.k[v];
        for(int v = 0; v < fm.n; ++v) {
            int m; in >> m;
            fm.parents[v].resize(m);
            for(int j = 0; j < m; ++j) in >> fm.parents[v][j];
            if(!std::is_sorted(fm.parents[v].begin(), fm.parents[v].end())) die("parents must ascend");
        }
        for(int v = 0; v < fm.n; ++v) {
            std::size_t sz = fm.k[v];
            for(int p : fm.parents[v]) sz *= fm.k[p];
            fm.cpt[v].resize(sz);
            for(auto& x : fm.cpt[v]) in >> x;
        }
        graph = build_graph(fm);
    }
    std::string mode;
    in >> mode;
    if(!in) die("truncated input");
    auto t1 = std::chrono::steady_clock::now();
    auto const vl = graph.vertex_list();

    if(mode == "bp") {
        double eps; int dump, ne;
        in >> eps >> dump >> ne;
        std::unordered_map<bn::vertex_type, bn::matrix_type> pre;
        for(int j = 0; j < ne; ++j) {
            int node, kv; in >> node >> kv;
            bn::matrix_type mt(1, kv, 0.0);
            for(int i = 0; i < kv; ++i) in >> mt[0][i];
            pre[vl[node]] = mt;
        }
        if(!in) die("truncated evidence");

        // ---- stepped run: the body of belief_propagation::operator() (hpp:31-159),
        //      one while-iteration at a time, calling the reference's own members.
        bn::inference::belief_propagation bp(graph);
        auto& g = bp.graph_;
        bp.initialize();
        for(auto const& node : g.vertex_list()) {
            bp.pi_[node].resize(1, node->selectable_num, 1.0);
            bp.lambda_[node].resize(1, node->selectable_num, 1.0);
            for(auto const& parent : g.in_vertexes(node)) bp.pi_i_[node][parent].resize(1, parent->selectable_num, 1.0);
            for(auto const& child : g.out_vertexes(node)) bp.lambda_k_[child][node].resize(1, node->selectable_num, 1.0);
            if(g.in_edges(node).empty()) {
                auto& pi = bp.pi_[node];
                auto& data = node->cpt[bn::condition_t()].second;
                pi.resize(1, node->selectable_num);
                pi.assign(data.cbegin(), data.cend());
            }
        }
        bp.preconditional_node_.clear();
        for(auto const& p : pre) {
            bp.preconditional_node_.push_back(p.first);
            bp.pi_[p.first] = bp.lambda_[p.first] = p.second;
        }
        std::vector<double> residuals;
        auto t2 = std::chrono::steady_clock::now();
        while(true) {
            for(auto const& node : g.vertex_list()) {
                for(auto const& parent : g.in_vertexes(node)) bp.calculate_pi_i(node, parent);
                for(auto const& child : g.out_vertexes(node)) bp.calculate_lambda_k(child, node);
            }
            for(auto const& node : g.vertex_list()) {
                if(bp.new_pi_.find(node) == bp.new_pi_.cend()) bp.calculate_pi(node);
                if(bp.new_lambda_.find(node) == bp.new_lambda_.cend()) bp.calculate_lambda(node);
            }
            double md = std::numeric_limits<double>::min();
            for(auto const& node : g.vertex_list()) {
                for(auto const& parent : g.in_vertexes(node))
                    for(std::size_t i = 0; i < parent->selectable_num; ++i)
                        md = std::max(md, std::abs(bp.new_pi_i_[node][parent][0][i] - bp.pi_i_[node][parent][0][i]));
                for(auto const& child : g.out_vertexes(node))
                    for(std::size_t i = 0; i < node->selectable_num; ++i)
                        md = std::max(md, std::abs(bp.new_lambda_k_[child][node][0][i] - bp.lambda_k_[child][node][0][i]));
            }
            for(auto const& o : bp.new_pi_) bp.pi_[o.first] = o.second;
            for(auto const& o : bp.new_lambda_) bp.lambda_[o.first] = o.second;
            for(auto const& o : bp.new_pi_i_) for(auto const& i : o.second) bp.pi_i_[o.first][i.first] = i.second;
            for(auto const& o : bp.new_lambda_k_) for(auto const& i : o.second) bp.lambda_k_[o.first][i.first] = i.second;
            bp.new_pi_.clear(); bp.new_lambda_.clear(); bp.new_pi_i_.clear(); bp.new_lambda_k_.clear();
            residuals.push_back(md);
            if(md < eps) break;
        }
        auto t3 = std::chrono::steady_clock::now();
        std::vector<std::vector<double>> bel(fm.n);
        for(int v = 0; v < fm.n; ++v) {
            auto raw = bp.pi_[vl[v]] % bp.lambda_[vl[v]];
            auto nb = bp.normalize(raw);
            bel[v] = nb[0];
        }

        // ---- plain call on a fresh instance: must be bit-identical
        bn::inference::belief_propagation bp2(graph);
        auto const res2 = bp2(pre, eps);
        bool same = true;
        for(int v = 0; v < fm.n; ++v) {
            auto const& m2 = res2.at(vl[v]);
            if(m2.height() != 1 || m2.width() != bel[v].size()) { same = false; continue; }
            for(std::size_t i = 0; i < bel[v].size(); ++i) {
                double a = m2[0][i], b = bel[v][i];
                if(!(a == b || (std::isnan(a) && std::isnan(b)))) same = false;
            }
        }

        std::printf("{\"mode\":\"bp\",\"sweeps\":%zu,\"stepped_equals_call\":%s,", residuals.size(), same ? "true" : "false");
        if(dsc_mode) {
            std::printf("\"flat\":{\"k\":[");
            for(int v = 0; v < fm.n; ++v) std::printf("%s%d", v ? "," : "", fm.k[v]);
            std::printf("],\"parents\":[");
            for(int v = 0; v < fm.n; ++v) {
                std::printf("%s[", v ? "," : "");
                for(std::size_t j = 0; j < fm.parents[v].size(); ++j) std::printf("%s%d", j ? "," : "", fm.parents[v][j]);
                std::printf("]");
            }
            std::printf("],\"cpt\":[");
            for(int v = 0; v < fm.n; ++v) { if(v) std::printf(","); print_vec(fm.cpt[v]); }
            std::printf("]},");
        }
        std::printf("\"build_s\":%.6f,\"sweep_s\":%.6f,", std::chrono::duration<double>(t1 - t0).count(),
                    std::chrono::duration<double>(t3 - t2).count());
        std::printf("\"residuals\":"); print_vec(residuals);
        std::printf(",\"beliefs\":[");
        for(int v = 0; v < fm.n; ++v) { if(v) std::printf(","); print_vec(bel[v]); }
        std::printf("]");
        if(dump) { // final messages in CSR edge order (child-major, parents ascending)
            std::printf(",\"pi_msg\":[");
            bool first = true;
            for(int v = 0; v < fm.n; ++v) for(int p : fm.parents[v]) {
                if(!first) std::printf(","); first = false;
                print_vec(bp.pi_i_[vl[v]][vl[p]][0]);
            }
            std::printf("],\"lambda_msg\":[");
            first = true;
            for(int v = 0; v < fm.n; ++v) for(int p : fm.parents[v]) {
                if(!first) std::printf(","); first = false;
                print_vec(bp.lambda_k_[vl[v]][vl[p]][0]);
            }
            std::printf("]");
        }
        std::printf("}\n");
        return same ? 0 : 3;
    }
    else if(mode == "lw") {
        unsigned long long ns; unsigned seed; int ne;
        in >> ns >> seed >> ne;
        bn::inference::likelihood_weighting::evidence_list ev;
        for(int j = 0; j < ne; ++j) { int node, st; in >> node >> st; ev[vl[node]] = st; }
        if(!in) die("truncated evidence");
        bn::inference::likelihood_weighting lw(graph);
        lw.probability_generator_.engine_.reset(new std::mt19937(seed)); // deterministic oracle
        auto t2 = std::chrono::steady_clock::now();
        auto const res = lw(ev, ns);
        auto t3 = std::chrono::steady_clock::now();
        std::printf("{\"mode\":\"lw\",\"samples\":%llu,\"seed\":%u,\"run_s\":%.6f,\"marginals\":[", ns, seed,
                    std::chrono::duration<double>(t3 - t2).count());
        for(int v = 0; v < fm.n; ++v) { if(v) std::printf(","); print_vec(res.at(vl[v])[0]); }
        std::printf("]}\n");
        return 0;
    }
    else if(mode == "lwms") {
        // likelihood_weighting::make_samples (hpp:62-117) with the engine reseeded: the joint-pattern
        // table, the marginals of the last unit and (from the table's total) the units executed
        unsigned long long unit; double eps; unsigned seed; int ne;
        in >> unit >> eps >> seed >> ne;
        bn::inference::likelihood_weighting::evidence_list ev;
        for(int j = 0; j < ne; ++j) { int node, st; in >> node >> st; ev[vl[node]] = st; }
        if(!in) die("truncated evidence");
        bn::inference::likelihood_weighting lw(graph);
        lw.probability_generator_.engine_.reset(new std::mt19937(seed));
        auto t2 = std::chrono::steady_clock::now();
        auto const res = lw.make_samples(ev, unit, eps);
        auto t3 = std::chrono::steady_clock::now();
        unsigned long long total = 0;
        for(auto const& p : res.first) total += p.second;
        std::printf("{\"mode\":\"lwms\",\"unit_size\":%llu,\"eps\":%.17g,\"seed\":%u,\"units\":%llu,\"run_s\":%.6f,\"marginals\":[",
                    unit, eps, seed, total / unit, std::chrono::duration<double>(t3 - t2).count());
        for(int v = 0; v < fm.n; ++v) { if(v) std::printf(","); print_vec(res.second.at(vl[v])[0]); }
        std::printf("],\"patterns\":[");
        bool first = true;
        for(auto const& p : res.first) {
            std::printf("%s[", first ? "" : ","); first = false;
            for(int v = 0; v < fm.n; ++v) std::printf("%d,", p.first.at(vl[v]));
            std::printf("%zu]", p.second);   // last entry = occurrence count
        }
        std::printf("]}\n");
        return 0;
    }
    else if(mode == "rs") {
        // rejection_sampling::operator()(condition, generate_sample_num) (hpp:33-62), engine reseeded
        int num; unsigned seed; int ne;
        in >> num >> seed >> ne;
        std::vector<std::pair<bn::vertex_type, int>> cond;
        for(int j = 0; j < ne; ++j) { int node, st; in >> node >> st; cond.emplace_back(vl[node], st); }
        if(!in) die("truncated condition");
        bn::inference::rejection_sampling rs(graph);
        rs.probability_generator_.engine_.reset(new std::mt19937(seed));
        auto t2 = std::chrono::steady_clock::now();
        auto const res = rs(cond, num);
        auto t3 = std::chrono::steady_clock::now();
        std::printf("{\"mode\":\"rs\",\"num\":%d,\"seed\":%u,\"run_s\":%.6f,\"marginals\":[", num, seed,
                    std::chrono::duration<double>(t3 - t2).count());
        for(int v = 0; v < fm.n; ++v) { if(v) std::printf(","); print_vec(res.at(vl[v])[0]); }
        std::printf("]}\n");
        return 0;
    }
    die("unknown mode");
    return 2;
}

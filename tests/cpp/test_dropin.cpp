// tests/cpp/test_dropin.cpp -- the reference's own BP test cases (libs/bayesian/test/
// belief_propagation.cpp: Pearl R,S->W,H parts 1-2, and the five "resume" chain cases) run through
// the drop-in class surface bn::inference::belief_propagation of this repository, plus a
// likelihood-weighting call.  Builds against EITHER data model:
//     -Iinclude -Iinclude/compat          (this repository's stand-in, GPU box)
//     -Iinclude -I/root/reference         (the reference's graph.hpp / matrix.hpp: true drop-in)
// Modes:  --flatten  print the flat model of both networks as JSON (no GPU needed)
//         --dsc F [--flatten]  load F with bn::serializer::dsc (the reference's loader or the compat one),
//                    print its flat model, and unless --flatten run BP (no evidence, eps 1e-3)
//         (default)  run inference, check the reference's teacher values, print 17-digit marginals
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include <bayesian/graph.hpp>
#include <bayesian/inference/belief_propagation.hpp>
#include <bayesian/inference/likelihood_weighting.hpp>
#include <bayesian/inference/rejection_sampling.hpp>
#include <bayesian/sampler.hpp>
#include <bayesian/serializer/dsc.hpp>

namespace {

struct node_spec {
    int arity;
    std::vector<int> parents;           // positions in the network
    std::vector<double> rows;           // row-major, first parent slowest
};

// table-driven construction through the public graph_t / cpt_t interface
bn::graph_t build(std::vector<node_spec> const& spec)
{
    bn::graph_t g;
    for(std::size_t i = 0; i < spec.size(); ++i)
    {
        auto v = g.add_vertex();
        v->id = static_cast<int>(i) + 1;
        v->selectable_num = spec[i].arity;
    }
    auto const vl = g.vertex_list();
    for(std::size_t i = 0; i < spec.size(); ++i)
        for(int p : spec[i].parents)
            if(!g.add_edge(vl[p], vl[i])) std::printf("add_edge failed\n");
    for(std::size_t i = 0; i < spec.size(); ++i)
    {
        std::vector<bn::vertex_type> ps;
        for(int p : spec[i].parents) ps.push_back(vl[p]);
        vl[i]->cpt.assign(ps, vl[i]);
        std::vector<int> st(ps.size(), 0);
        std::size_t const k = spec[i].arity;
        for(std::size_t r = 0; r * k < spec[i].rows.size(); ++r)
        {
            bn::condition_t cond;
            for(std::size_t j = 0; j < ps.size(); ++j) cond[ps[j]] = st[j];
            vl[i]->cpt[cond].second.assign(spec[i].rows.begin() + r * k, spec[i].rows.begin() + (r + 1) * k);
            for(std::size_t j = ps.size(); j-- > 0;)
            {
                if(++st[j] < spec[spec[i].parents[j]].arity) break;
                st[j] = 0;
            }
        }
    }
    return g;
}

std::vector<node_spec> pearl_spec()
{
    return {{2, {}, {0.2, 0.8}},
            {2, {}, {0.1, 0.9}},
            {2, {0}, {1.0, 0.0, 0.2, 0.8}},
            {2, {0, 1}, {1.0, 0.0, 1.0, 0.0, 0.9, 0.1, 0.0, 1.0}}};
}

std::vector<node_spec> resume_spec()
{
    return {{3, {}, {0.30, 0.60, 0.10}},
            {3, {0}, {0.20, 0.30, 0.50, 0.30, 0.30, 0.40, 0.80, 0.10, 0.10}},
            {2, {1}, {0.50, 0.50, 0.70, 0.30, 0.40, 0.60}},
            {3, {2}, {0.40, 0.30, 0.30, 0.20, 0.60, 0.20}}};
}

int failures = 0;

void close_pct(double value, double teacher, double pct, char const* what)
{
    bool ok = (teacher == 0.0) ? std::fabs(value) < 1e-12 : std::fabs(value - teacher) / std::fabs(teacher) * 100.0 <= pct;
    if(!ok) { ++failures; std::printf("FAIL %s: %.17g vs teacher %.17g (%g %%)\n", what, value, teacher, pct); }
}

bn::matrix_type one_hot(std::vector<double> const& v)
{
    bn::matrix_type m(1, v.size());
    m[0] = v;
    return m;
}

template<class V> void print_json_array(char const* name, V const& v, bool last = false)
{
    std::printf("\"%s\":[", name);
    for(std::size_t i = 0; i < v.size(); ++i) std::printf("%s%.17g", i ? "," : "", static_cast<double>(v[i]));
    std::printf("]%s", last ? "" : ",");
}

void print_flat(char const* name, bn::graph_t const& g, bool last)
{
    auto const fm = bn::mi355x::flatten(g);
    std::printf("\"%s\":{", name);
    print_json_array("k", fm.k);
    print_json_array("in_ptr", fm.in_ptr);
    print_json_array("in_idx", fm.in_idx);
    print_json_array("cpt_off", fm.cpt_off);
    print_json_array("cpt", fm.cpt, true);
    std::printf("}%s", last ? "" : ",");
}

void print_marginals(char const* name, bn::graph_t const& g,
                     std::unordered_map<bn::vertex_type, bn::matrix_type> const& res, bool last = false)
{
    std::printf("\"%s\":[", name);
    auto const vl = g.vertex_list();
    bool first = true;
    for(auto const& v : vl)
        for(double x : res.at(v)[0]) { std::printf("%s%.17g", first ? "" : ",", x); first = false; }
    std::printf("]%s", last ? "" : ",");
}

// run()'s non-owning view against operator()'s map: same vertices, same arities, same bits
template<class View, class Map>
void view_equals_map(View const& view, Map const& map, bn::graph_t const& g, char const* what)
{
    auto const vl = g.vertex_list();
    bool ok = view.size() == map.size() && view.size() == vl.size();
    std::size_t i = 0, doubles = 0;
    for(auto const& e : view)
    {
        if(i >= vl.size() || e.vertex != vl[i]) { ok = false; break; }   // iteration is in vertex_list() order
        auto const it = map.find(e.vertex);
        if(it == map.end() || it->second.height() != 1 || it->second.width() != e.k || view.k(e.vertex) != e.k || view[e.vertex] != e.p)
        { ok = false; break; }
        for(std::size_t j = 0; j < e.k; ++j)
            if(std::memcmp(&it->second[0][j], e.p + j, sizeof(double)) != 0) ok = false;
        auto const cell = view.matrix(e.vertex);
        for(std::size_t j = 0; j < e.k; ++j)
            if(std::memcmp(&cell[0][j], e.p + j, sizeof(double)) != 0) ok = false;
        doubles += e.k;
        ++i;
    }
    if(doubles != view.doubles() || view.data() != (vl.empty() ? view.data() : view[vl[0]])) ok = false;
    bool threw = false;
    try { view[std::make_shared<bn::vertex_t>()]; } catch(std::out_of_range const&) { threw = true; }
    if(!ok || !threw) { ++failures; std::printf("FAIL %s: marginals_view differs from the returned map\n", what); }
}

} // namespace

int main(int argc, char** argv)
{
    bn::graph_t const pearl = build(pearl_spec());
    bn::graph_t const chain = build(resume_spec());
    if(argc > 1 && std::strcmp(argv[1], "--flatten") == 0)
    {
        std::printf("{");
        print_flat("pearl", pearl, false);
        print_flat("resume_chain", chain, true);
        std::printf("}\n");
        return 0;
    }

    if(argc > 2 && std::strcmp(argv[1], "--dsc") == 0)
    {
        bn::graph_t const net = bn::serializer::dsc().from_file(argv[2]);
        std::printf("{");
        bool const only_flat = argc > 3 && std::strcmp(argv[3], "--flatten") == 0;
        print_flat("net", net, only_flat);
        if(!only_flat)
        {
            bn::inference::belief_propagation bp(net);
            auto const res = bp();
            std::printf("\"sweeps\":%d,", bp.last_sweeps());
            print_marginals("beliefs", net, res, true);
        }
        std::printf("}\n");
        return 0;
    }

    std::printf("{");
    {   // belief_propagation_pearl_part1 / part2
        auto const v = pearl.vertex_list();
        bn::inference::belief_propagation bp(pearl);
        auto const r1 = bp();
        double const t1[4][2] = {{.2, .8}, {.1, .9}, {.36, .64}, {.272, .728}};
        for(int i = 0; i < 4; ++i)
            for(int j = 0; j < 2; ++j) close_pct(r1.at(v[i])[0][j], t1[i][j], 0.01, "pearl part1");
        print_marginals("pearl_part1", pearl, r1);
        std::unordered_map<bn::vertex_type, bn::matrix_type> pre;
        pre[v[3]] = one_hot({1, 0});
        auto const r2 = bp(pre);
        double const t2[4][2] = {{.7353, .2647}, {.3382, .6618}, {.7882, .2118}, {1.0, 0.0}};
        for(int i = 0; i < 4; ++i)
            for(int j = 0; j < 2; ++j) close_pct(r2.at(v[i])[0][j], t2[i][j], 0.1, "pearl part2");
        print_marginals("pearl_part2", pearl, r2);
        std::printf("\"pearl_part2_sweeps\":%d,", bp.last_sweeps());
        view_equals_map(bp.run(pre), r2, pearl, "belief_propagation pearl part2");
        view_equals_map(bp.run(bp.prepare(pre)), r2, pearl, "belief_propagation pearl part2, prepared evidence");
        view_equals_map(bp.run(), r1, pearl, "belief_propagation pearl part1");
    }
    {   // belief_propagation_resume_ex, _sample1 .. _sample4  (3 % tolerance, one queried node each)
        auto const v = chain.vertex_list();
        struct rc { std::vector<std::pair<int, std::vector<double>>> ev; int query; std::vector<double> teacher; };
        std::vector<rc> const cases = {
            {{{1, {0, 0, 1}}, {3, {1, 0, 0}}}, 2, {0.570, 0.430}},
            {{{2, {0, 1}}}, 1, {0.330, 0.170, 0.500}},
            {{{0, {0, 1, 0}}, {2, {0, 1}}}, 1, {0.310, 0.190, 0.500}},
            {{{3, {0, 0, 1}}}, 0, {0.300, 0.600, 0.100}},
            {{{0, {1, 0, 0}}}, 1, {0.200, 0.300, 0.500}},
        };
        int idx = 0;
        for(auto const& c : cases)
        {
            std::unordered_map<bn::vertex_type, bn::matrix_type> pre;
            for(auto const& e : c.ev) pre[v[e.first]] = one_hot(e.second);
            bn::inference::belief_propagation func(chain);
            auto const res = func(pre);
            for(std::size_t j = 0; j < c.teacher.size(); ++j) close_pct(res.at(v[c.query])[0][j], c.teacher[j], 3.0, "resume");
            print_marginals(("resume_" + std::to_string(idx++)).c_str(), chain, res);
            view_equals_map(func.run(pre), res, chain, "belief_propagation resume");
        }
        // the same five queries in ONE call (run_batch, an extension): bit-for-bit what the single calls return
        std::vector<std::unordered_map<bn::vertex_type, bn::matrix_type>> queries;
        for(auto const& c : cases)
        {
            std::unordered_map<bn::vertex_type, bn::matrix_type> pre;
            for(auto const& e : c.ev) pre[v[e.first]] = one_hot(e.second);
            queries.push_back(pre);
        }
        bn::inference::belief_propagation func(chain);
        auto const batch = func.run_batch(queries);
        if(batch.size() != queries.size()) { ++failures; std::printf("FAIL run_batch size\n"); }
        for(std::size_t q = 0; q < batch.size(); ++q)
        {
            auto const single = func(queries[q]);
            for(auto const& node : v)
                for(std::size_t j = 0; j < single.at(node).width(); ++j)
                    if(!(batch[q].at(node)[0][j] == single.at(node)[0][j])) { ++failures; std::printf("FAIL run_batch differs from operator()\n"); }
        }
    }
    {   // likelihood weighting on Pearl, H = 0: within 2 % of the exact marginals at 4e5 samples
        auto const v = pearl.vertex_list();
        bn::inference::likelihood_weighting lw(pearl);
        lw.seed(2024);
        bn::inference::likelihood_weighting::evidence_list ev;
        ev[v[3]] = 0;
        auto const res = lw(ev, 400000);
        double const exact[3][2] = {{0.73529411764705888, 0.26470588235294118}, {0.33823529411764708, 0.66176470588235292},
                                    {0.78823529411764715, 0.21176470588235297}};
        for(int i = 0; i < 3; ++i)
            for(int j = 0; j < 2; ++j) close_pct(res.at(v[i])[0][j], exact[i][j], 2.0, "lw pearl");
        print_marginals("lw_pearl", pearl, res);
        lw.seed(2024);   // the same samples again: the fp64 atomics of the histogram may round differently, the view reads what this call made
        auto const view = lw.run(ev, 400000);
        for(int i = 0; i < 3; ++i)
            for(int j = 0; j < 2; ++j) close_pct(view[v[i]][j], res.at(v[i])[0][j], 1e-7, "lw run() vs operator()");
        view_equals_map(view, view.to_map(), pearl, "likelihood_weighting");
    }
    {   // make_samples (reference likelihood_weighting.hpp:62-117) on Pearl, H = 0
        auto const v = pearl.vertex_list();
        bn::inference::likelihood_weighting lw(pearl);
        lw.seed(7);
        bn::inference::likelihood_weighting::evidence_list ev;
        ev[v[3]] = 0;
        auto const made = lw.make_samples(ev, 200000, 0.005);
        std::size_t total = 0;
        bool consistent = true;
        for(auto const& p : made.first)
        {
            total += p.second;
            if(p.first.size() != 4 || p.first.at(v[3]) != 0) consistent = false;   // evidence is clamped
        }
        if(total == 0 || total % 200000 != 0 || total < 400000 || !consistent || made.first.size() > 8)
        { ++failures; std::printf("FAIL make_samples: %zu samples, %zu patterns\n", total, made.first.size()); }
        double const exact[3][2] = {{0.73529411764705888, 0.26470588235294118}, {0.33823529411764708, 0.66176470588235292},
                                    {0.78823529411764715, 0.21176470588235297}};
        for(int i = 0; i < 3; ++i)
            for(int j = 0; j < 2; ++j) close_pct(made.second.at(v[i])[0][j], exact[i][j], 2.0, "make_samples pearl");
        std::printf("\"make_samples_total\":%zu,\"make_samples_patterns\":%zu,", total, made.first.size());
        // units executed, the joint-pattern table and the marginals: compared by the python side with
        // an oracle run of the same loop fed with the GPU's own stream (oracle/ref_replay.c)
        std::printf("\"make_samples_units\":%llu,\"make_samples_table\":[", static_cast<unsigned long long>(lw.last_units()));
        bool first_row = true;
        for(auto const& p : made.first)
        {
            std::printf("%s[", first_row ? "" : ","); first_row = false;
            for(std::size_t i = 0; i < v.size(); ++i) std::printf("%d,", static_cast<int>(p.first.at(v[i])));
            std::printf("%zu]", p.second);
        }
        std::printf("],");
        print_marginals("make_samples_marginals", pearl, made.second);
    }
    {   // rejection_sampling_standard (libs/bayesian/test/rejection_sampling.cpp): 5-node net,
        // condition {v4 = 1, v1 = 0}, P(v2) ~ {.62, .38} within 10 %
        std::vector<node_spec> const spec = {{2, {}, {0.5, 0.5}},
                                             {2, {0}, {0.8, 0.2, 0.1, 0.9}},
                                             {2, {0}, {0.7, 0.3, 0.4, 0.6}},
                                             {2, {1}, {0.6, 0.4, 0.1, 0.9}},
                                             {2, {1, 2}, {0.1, 0.9, 0.2, 0.8, 0.3, 0.7, 0.4, 0.6}}};
        bn::graph_t const net = build(spec);
        auto const v = net.vertex_list();
        bn::inference::rejection_sampling func(net);
        func.seed(99);
        auto const result = func({{v[3], 1}, {v[0], 0}});
        close_pct(result.at(v[1])[0][0], 0.62, 10, "rejection v2[0]");
        close_pct(result.at(v[1])[0][1], 0.38, 10, "rejection v2[1]");
        close_pct(result.at(v[3])[0][1], 1.0, 1e-9, "rejection keeps the condition");
        print_marginals("rejection", net, result);
        func.seed(99);   // integer counts: the view of the same draws is the map bit for bit
        view_equals_map(func.run({{v[3], 1}, {v[0], 0}}), result, net, "rejection_sampling");
        std::printf("\"rejection_drawn\":%llu,", static_cast<unsigned long long>(func.last_drawn()));
    }
    {   // CPT liveness.  The reference reads node->cpt at every call (belief_propagation.hpp:61, :186, :252): an edited table is
        // seen by the next operator().  The drop-in flattens the tables once, in its constructor: WITHOUT reload() the next call
        // still answers for the old tables (documented difference), WITH it the functor equals one built from the edited network.
        bn::graph_t const net = build(resume_spec());
        auto const v = net.vertex_list();
        std::unordered_map<bn::vertex_type, bn::matrix_type> pre;
        pre[v[3]] = one_hot({0, 0, 1});
        bn::inference::belief_propagation bp(net);
        bn::inference::likelihood_weighting lw(net);
        lw.seed(5);
        bn::inference::likelihood_weighting::evidence_list lev;
        lev[v[3]] = 2;
        print_marginals("reload_before", net, bp(pre));
        auto const lw_before = lw(lev, 20000);
        bn::condition_t cond;
        cond[v[1]] = 1;
        std::vector<double> const row = {0.1, 0.9};          // P(C | B = 1): was {0.7, 0.3}
        v[2]->cpt[cond].second.assign(row.begin(), row.end());
        print_marginals("reload_stale", net, bp(pre));
        bp.reload();
        print_marginals("reload_fresh", net, bp(pre));
        bn::inference::belief_propagation rebuilt(net);
        print_marginals("reload_rebuilt", net, rebuilt(pre));
        lw.reload();
        lw.seed(5);
        auto const lw_fresh = lw(lev, 20000);
        bn::inference::likelihood_weighting lw2(net);
        lw2.seed(5);
        auto const lw_rebuilt = lw2(lev, 20000);
        bool same = true, moved = false;
        for(auto const& node : v)
            for(std::size_t j = 0; j < lw_fresh.at(node).width(); ++j)
            {
                // (the weighted histogram is summed with fp64 atomics: equal to rounding, not to the bit)
                if(std::fabs(lw_fresh.at(node)[0][j] - lw_rebuilt.at(node)[0][j]) > 1e-9) same = false;
                if(std::fabs(lw_fresh.at(node)[0][j] - lw_before.at(node)[0][j]) > 0.02) moved = true;
            }
        if(!same || !moved) { ++failures; std::printf("FAIL likelihood_weighting::reload (same %d, moved %d)\n", int(same), int(moved)); }
        // a changed STRUCTURE is refused
        bn::graph_t other = build(pearl_spec());
        bool threw = false;
        try { bp.reload(other); } catch(std::runtime_error const&) { threw = true; }
        if(!threw) { ++failures; std::printf("FAIL reload accepted another structure\n"); }
    }
    {   // sampler::make_cpt (reference sampler.hpp:81-163): the pattern table make_samples returns,
        // loaded into bn::sampler, refits the CPTs of a structure-only copy of the network
        auto const v = pearl.vertex_list();
        bn::inference::likelihood_weighting lw(pearl);
        lw.seed(11);
        auto const made = lw.make_samples({}, 300000, 0.01);
        // the table is keyed by pearl's vertices; the structure-only copy shares them through a
        // second graph over the same vertex objects is not possible, so refit pearl itself on a
        // saved copy of its CPT rows
        auto const before = bn::mi355x::flatten(pearl);
        bn::sampler smp;
        if(smp.make_cpt(pearl)) { ++failures; std::printf("FAIL make_cpt without samples must return false\n"); }
        smp.load_sample(made.first);
        std::size_t total = 0;
        for(auto const& p : made.first) total += p.second;
        if(smp.sampling_size() != total) { ++failures; std::printf("FAIL sampling_size\n"); }
        if(!smp.make_cpt(pearl)) { ++failures; std::printf("FAIL make_cpt returned false\n"); }
        auto const after = bn::mi355x::flatten(pearl);
        // rows of W given R=0 and H given (R=0, S=*) are deterministic in the generating CPT;
        // every row has >= 4 % of the samples, so 0.01 absolute is > 5 sigma
        for(std::size_t i = 0; i < before.cpt.size(); ++i)
            if(std::fabs(before.cpt[i] - after.cpt[i]) > 0.01)
            { ++failures; std::printf("FAIL make_cpt entry %zu: %.6f vs %.6f\n", i, after.cpt[i], before.cpt[i]); }
        print_json_array("fitted_cpt", after.cpt);
        // exact host-side recount of the same table
        std::vector<double> cnt(before.cpt.size(), 0.0);
        for(auto const& p : made.first)
            for(std::size_t i = 0; i < v.size(); ++i)
            {
                std::size_t row = 0;
                for(std::int32_t e = before.in_ptr[i]; e < before.in_ptr[i + 1]; ++e)
                    row = row * before.k[before.in_idx[e]] + p.first.at(v[before.in_idx[e]]);
                cnt[before.cpt_off[i] + row * before.k[i] + p.first.at(v[i])] += static_cast<double>(p.second);
            }
        print_json_array("fit_counts", cnt, true);
    }
    std::printf("}\n");
    if(failures) std::printf("%d FAILURES\n", failures);
    return failures ? 1 : 0;
}

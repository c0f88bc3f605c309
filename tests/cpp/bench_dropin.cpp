// tests/cpp/bench_dropin.cpp -- what a user of the reference's API pays per query: the time of
// bn::inference::belief_propagation::operator()(precondition, epsilon) (reference
// belief_propagation.hpp:31-159) measured AT THE CLASS SURFACE, on BASELINE.json configs[0], [1], [2],
// split into its parts:
//     marshal      evidence map -> flat arrays (hash lookups vertex -> position)
//     c_abi        bn_bp_run_view: evidence H2D, run to convergence, marginals D2H, one synchronisation
//     map_build    the reference's return type, unordered_map<vertex_type, matrix_type> (:14, :151-158)
//     map_destroy  the caller dropping that map
// beside run(), the same query read through bn::mi355x::marginals_view (no map).
//
// The networks are built through graph_t / cpt_t like any user's (include/compat: the reference's graph_t is a
// dense V x V matrix, 160 GB at 10^5 nodes) with the generators of bayesiannetwork_amd/synth.py restated here
// (splitmix64 streams), so bench.py can check sweeps and a checksum of the marginals against its own run of
// the same network and evidence through the C ABI.
//
//     bench_dropin [--configs alarm,dag,grid] [--dsc FILE] [--reps N] [--grid ROWS] [--checksum]
// --checksum: no GPU call; prints checksums of the flat models and of query 0's evidence (CPU test against synth.py).
// One JSON line on stdout.  Built by __graft_entry__.build() with plain g++; needs a GPU to run.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include <bayesian/graph.hpp>
#include <bayesian/inference/belief_propagation.hpp>
#include <bayesian/serializer/dsc.hpp>

namespace {

typedef std::unordered_map<bn::vertex_type, bn::matrix_type> evidence_map;

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// word j of stream `seed` (bayesiannetwork_amd/synth.py: splitmix64)
std::uint64_t splitmix64(std::uint64_t const seed, std::uint64_t const j)
{
    std::uint64_t z = seed + (j + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
double uniform01(std::uint64_t const seed, std::uint64_t const j)
{
    return static_cast<double>(splitmix64(seed, j) >> 11) * (1.0 / 9007199254740992.0);
}

struct structure {
    std::vector<int> k;
    std::vector<std::vector<int>> parents;   // ascending
};

// synth.grid: node (r,c) <- (r-1,c), (r,c-1)
structure grid_structure(int const rows, int const cols, int const k)
{
    structure s;
    s.k.assign(static_cast<std::size_t>(rows) * cols, k);
    s.parents.resize(s.k.size());
    for(int r = 0; r < rows; ++r)
        for(int c = 0; c < cols; ++c)
        {
            auto& p = s.parents[static_cast<std::size_t>(r) * cols + c];
            if(r > 0) p.push_back((r - 1) * cols + c);
            if(c > 0) p.push_back(r * cols + c - 1);
        }
    return s;
}

// synth.random_dag: node i draws m in {0..min(max_parents, i - lo)} parents without replacement from [lo, i)
structure dag_structure(int const n, int const max_parents, int const window, int const k, std::uint64_t const seed)
{
    structure s;
    s.k.assign(n, k);
    s.parents.resize(n);
    std::uint64_t const stream = seed ^ 0x5DEECE66Dull;
    std::uint64_t di = 0;
    for(int i = 0; i < n; ++i)
    {
        int const lo = std::max(0, i - window);
        int const cap = std::min(max_parents, i - lo);
        std::size_t const m = static_cast<std::size_t>(splitmix64(stream, di++) % static_cast<std::uint64_t>(cap + 1));
        std::set<int> chosen;
        while(chosen.size() < m) chosen.insert(lo + static_cast<int>(splitmix64(stream, di++) % static_cast<std::uint64_t>(i - lo)));
        s.parents[i].assign(chosen.begin(), chosen.end());
    }
    return s;
}

// graph_t + CPTs through the public interface; rows 0.1 + 0.9 u divided by the row sum (synth._random_cpts: the sum
// of a row of kk entries is ((r1 + r2) + ... + r_{kk-1}) + r0, numpy's add.reduceat order)
bn::graph_t build_graph(structure const& s, std::uint64_t const seed)
{
    bn::graph_t g;
    std::size_t const n = s.k.size();
    for(std::size_t i = 0; i < n; ++i)
    {
        auto v = g.add_vertex();
        v->id = static_cast<int>(i);
        v->selectable_num = static_cast<std::size_t>(s.k[i]);
    }
    auto const& vl = g.vertex_list();
    // every in-edge of v before v has a child: add_edge's cycle check then starts from a childless vertex
    for(std::size_t i = 0; i < n; ++i)
        for(int p : s.parents[i])
            if(!g.add_edge(vl[p], vl[i])) { std::fprintf(stderr, "add_edge failed\n"); std::exit(2); }
    std::uint64_t word = 0;
    for(std::size_t i = 0; i < n; ++i)
    {
        std::vector<bn::vertex_type> ps;
        for(int p : s.parents[i]) ps.push_back(vl[p]);
        vl[i]->cpt.assign(ps, vl[i]);
        std::vector<int> st(ps.size(), 0);
        std::size_t rows = 1;
        for(int p : s.parents[i]) rows *= static_cast<std::size_t>(s.k[p]);
        std::size_t const kk = static_cast<std::size_t>(s.k[i]);
        std::vector<double> r(kk);
        for(std::size_t row = 0; row < rows; ++row)
        {
            for(std::size_t j = 0; j < kk; ++j) r[j] = 0.1 + 0.9 * uniform01(seed, word++);
            double sum = r[kk > 1 ? 1 : 0];
            for(std::size_t j = 2; j < kk; ++j) sum += r[j];
            if(kk > 1) sum += r[0];
            for(std::size_t j = 0; j < kk; ++j) r[j] /= sum;
            bn::condition_t cond;
            for(std::size_t j = 0; j < ps.size(); ++j) cond[ps[j]] = st[j];
            vl[i]->cpt[cond].second.assign(r.begin(), r.end());
            for(std::size_t j = ps.size(); j-- > 0;)
            {
                if(++st[j] < s.k[s.parents[i][j]]) break;
                st[j] = 0;
            }
        }
    }
    return g;
}

// synth.random_evidence: floor(frac * V) distinct nodes by repeated x mod V, state x' mod k, one-hot vectors
evidence_map random_evidence(bn::graph_t const& g, double const frac, std::uint64_t const seed)
{
    auto const& vl = g.vertex_list();
    std::size_t const want = static_cast<std::size_t>(frac * static_cast<double>(vl.size()));
    std::uint64_t const stream = seed ^ 0xE71DE9CEull;
    std::unordered_map<std::size_t, int> chosen;
    for(std::uint64_t i = 0; chosen.size() < want; i += 2)
    {
        std::size_t const v = static_cast<std::size_t>(splitmix64(stream, i) % vl.size());
        if(!chosen.count(v)) chosen[v] = static_cast<int>(splitmix64(stream, i + 1) % vl[v]->selectable_num);
    }
    evidence_map ev;
    for(auto const& c : chosen)
    {
        bn::matrix_type m(1, vl[c.first]->selectable_num, 0.0);
        m[0][c.second] = 1.0;
        ev.emplace(vl[c.first], m);
    }
    return ev;
}

// order-sensitive checksum that numpy restates in one line: sum_i word_i * (2 i + 1) mod 2^64
// (doubles as their 64-bit patterns, int32 values widened)
template<class T> std::uint64_t wsum64(T const* data, std::size_t const count)
{
    std::uint64_t h = 0;
    for(std::size_t i = 0; i < count; ++i)
    {
        std::uint64_t w;
        if(sizeof(T) == 8) std::memcpy(&w, data + i, 8);
        else w = static_cast<std::uint64_t>(static_cast<std::int64_t>(data[i]));
        h += w * (2 * static_cast<std::uint64_t>(i) + 1);
    }
    return h;
}

double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

struct flat_evidence {
    std::vector<std::int32_t> node, off;
    std::vector<double> val;
};

bool checksum_only = false;

// --checksum (no GPU): the flat model and the evidence of query 0, for comparison with bayesiannetwork_amd/synth.py
void print_checksums(char const* name, bn::graph_t const& g, double const ev_frac, bool const last)
{
    double const t_flat = now_ms();
    auto const fm = bn::mi355x::flatten(g);
    double const flatten_ms = now_ms() - t_flat;
    auto const ev = random_evidence(g, ev_frac, 7);
    std::vector<std::int32_t> pairs;   // (node, state) ascending by node
    for(auto const& p : ev)
    {
        std::int32_t st = 0;
        for(std::size_t j = 0; j < p.second.width(); ++j) if(p.second[0][j] == 1.0) st = static_cast<std::int32_t>(j);
        pairs.push_back(fm.index.at(p.first) * 256 + st);
    }
    std::sort(pairs.begin(), pairs.end());
    std::printf("\"%s\":{\"nodes\":%zu,\"edges\":%zu,\"flatten_ms\":%.3f,\"in_idx\":\"%016llx\",\"cpt\":\"%016llx\",\"evidence\":\"%016llx\"}%s", name,
                fm.k.size(), fm.in_idx.size(), flatten_ms,
                static_cast<unsigned long long>(wsum64(fm.in_idx.data(), fm.in_idx.size())),
                static_cast<unsigned long long>(wsum64(fm.cpt.data(), fm.cpt.size())),
                static_cast<unsigned long long>(wsum64(pairs.data(), pairs.size())), last ? "" : ",");
}

void bench_network(char const* name, bn::graph_t const& g, double const graph_build_ms, double const ev_frac, double const eps,
                   int const reps, bool const last)
{
    if(checksum_only) { print_checksums(name, g, ev_frac, last); return; }
    std::size_t const n = g.vertex_list().size();
    double t0 = now_ms();
    bn::inference::belief_propagation bp(g);
    double const construct_ms = now_ms() - t0;

    std::vector<evidence_map> evs;
    for(int q = 0; q < 8; ++q) evs.push_back(random_evidence(g, ev_frac, 7 + q));

    // what the constructor did, split (reference belief_propagation.hpp:16-19 copies a graph): flattening graph_t through its public
    // interface, then bn_create -- host planning (the layout of the default path; the others' come with their first use) and the
    // device side (allocations, uploads).  This second engine on the same flat model is a steady-state construction: the runtime,
    // the code objects and the allocator's pools exist by now.
    t0 = now_ms();
    auto const fm = bn::mi355x::flatten(g);
    double const flatten_ms = now_ms() - t0;
    t0 = now_ms();
    bn::mi355x::engine_handle raw(fm);
    double const create_ms = now_ms() - t0;
    double create_host_ms = 0.0, create_device_ms = 0.0;
    {
        char const* const host_keys[] = {"create_us_plan", "create_us_small", "create_us_mid", "create_us_dag"};
        for(char const* k : host_keys) create_host_ms += static_cast<double>(bn_get_info(raw.get(), k)) * 1e-3;
        create_device_ms = static_cast<double>(bn_get_info(raw.get(), "create_us_device")) * 1e-3;
    }
    double construct_again_ms = 0.0;
    {   // ... and a second functor, whole (flatten + bn_create + the graph copy), destroyed again outside the clock
        t0 = now_ms();
        bn::inference::belief_propagation again(g);
        construct_again_ms = now_ms() - t0;
    }
    std::vector<flat_evidence> flat(evs.size());
    for(std::size_t q = 0; q < evs.size(); ++q)
    {
        flat[q].off.push_back(0);
        for(auto const& p : evs[q])
        {
            flat[q].node.push_back(fm.index.at(p.first));
            flat[q].val.insert(flat[q].val.end(), p.second[0].begin(), p.second[0].end());
            flat[q].off.push_back(static_cast<std::int32_t>(flat[q].val.size()));
        }
    }
    auto run_raw = [&](std::size_t q) {
        double const* b = nullptr;
        std::int32_t sw = 0;
        double res = 0;
        bn::mi355x::engine_handle::check(bn_bp_run_view(raw.get(), static_cast<std::int32_t>(flat[q].node.size()), flat[q].node.data(),
                                                        flat[q].off.data(), flat[q].val.data(), eps, 0, &b, &sw, &res));
        return sw;
    };

    // warm-up by time: code objects, clocks, the allocator's arenas
    t0 = now_ms();
    for(int i = 0; i < 2 || (now_ms() - t0 < 150.0 && i < 4096); ++i)
    {
        bp.run(evs[i % evs.size()], eps);
        run_raw(i % evs.size());
    }
    { auto warm = bp(evs[0], eps); }

    // checksum of query 0: the flat marginals the view reads == what the map holds
    auto const v0 = bp.run(evs[0], eps);
    std::uint64_t const sum_view = wsum64(v0.data(), v0.doubles());
    int const sweeps0 = bp.last_sweeps();
    bool map_equals_view = true;
    {
        auto const m0 = bp(evs[0], eps);
        auto const v1 = bp.run(evs[0], eps);
        if(m0.size() != n) map_equals_view = false;
        for(auto const& e : v1)
        {
            auto const it = m0.find(e.vertex);
            if(it == m0.end() || it->second.height() != 1 || it->second.width() != e.k) { map_equals_view = false; continue; }
            if(std::memcmp(it->second[0].data(), e.p, e.k * sizeof(double)) != 0) map_equals_view = false;
        }
    }

    // Two passes over the same cycle of queries.  First what a caller of run() executes -- and, beside it, the bare C ABI call on
    // arrays marshalled beforehand -- with nothing else in between: a caller that reads marginals in place builds no maps, and
    // building + dropping 10^5 map entries between two queries would evict the evidence maps from the caches.
    std::vector<double> t_raw, t_run, t_build, t_destroy, t_op, t_old;
    long sweeps = 0;
    for(int i = 0; i < reps; ++i)
    {
        std::size_t const q = static_cast<std::size_t>(i) % evs.size();
        double a = now_ms();
        sweeps += run_raw(q);
        double b = now_ms();
        t_raw.push_back(b - a);
    }
    double sink = 0;
    for(int i = 0; i < reps; ++i)
    {
        std::size_t const q = static_cast<std::size_t>(i) % evs.size();
        double const a = now_ms();
        auto const view = bp.run(evs[q], eps);
        double const b = now_ms();
        t_run.push_back(b - a);
        sink += view.data()[0];
    }
    // ... and run() on evidence prepared once (bn::inference::belief_propagation::prepare): no walk over the caller's map
    std::vector<bn::inference::belief_propagation::evidence_arrays> prepared;
    for(auto const& e : evs) prepared.push_back(bp.prepare(e));
    std::vector<double> t_prep;
    for(int i = 0; i < reps; ++i)
    {
        std::size_t const q = static_cast<std::size_t>(i) % evs.size();
        double const a = now_ms();
        auto const view = bp.run(prepared[q], eps);
        double const b = now_ms();
        t_prep.push_back(b - a);
        sink += view.data()[0];
    }
    if(sink < 0) std::printf(" ");
    // ... then the reference's return type: built from the view, dropped, and the whole operator() as the reference's user calls it
    for(int i = 0; i < reps; ++i)
    {
        std::size_t const q = static_cast<std::size_t>(i) % evs.size();
        auto const view = bp.run(evs[q], eps);
        double a, b;
        {
            a = now_ms();
            auto* map = new bn::inference::belief_propagation::return_type(view.to_map());
            b = now_ms();
            t_build.push_back(b - a);
            a = now_ms();
            delete map;
            b = now_ms();
            t_destroy.push_back(b - a);
        }
        {   // the class surface: result built, used, dropped
            a = now_ms();
            {
                auto const result = bp(evs[q], eps);
                if(result.size() != n) std::exit(3);
            }
            b = now_ms();
            t_op.push_back(b - a);
        }
        if(i < std::max(3, reps / 8))
        {   // how the map was built before (a default-constructed entry + a copy assignment per node): for the cost table
            auto const again = bp.run(evs[q], eps);
            a = now_ms();
            {
                bn::inference::belief_propagation::return_type result;
                for(auto const& e : again)
                {
                    bn::matrix_type m(1, e.k);
                    m.assign(e.begin(), e.end());
                    result[e.vertex] = m;
                }
                b = now_ms();
                t_old.push_back(b - a);
            }
        }
    }
    double const raw_ms = median(t_raw), run_ms = median(t_run);
    std::printf("\"%s\":{\"nodes\":%zu,\"edges\":%zu,\"evidence_nodes\":%zu,\"eps\":%g,\"reps\":%d,\"sweeps_per_query\":%.3f,"
                "\"graph_build_ms\":%.3f,\"functor_construct_ms\":%.3f,\"functor_construct_again_ms\":%.3f,\"flatten_ms\":%.3f,\"bn_create_ms\":%.3f,"
                "\"bn_create_host_ms\":%.3f,\"bn_create_device_ms\":%.3f,"
                "\"run_view_ms\":%.5f,\"run_prepared_ms\":%.5f,\"c_abi_ms\":%.5f,\"marshal_ms\":%.5f,\"map_build_ms\":%.5f,\"map_destroy_ms\":%.5f,"
                "\"operator_ms\":%.5f,\"map_build_copy_assign_ms\":%.5f,"
                "\"sweeps_query0\":%d,\"wsum64_query0\":\"%016llx\",\"map_equals_view\":%s}%s",
                name, n, g.edge_list().size(), evs[0].size(), eps, reps, static_cast<double>(sweeps) / reps,
                graph_build_ms, construct_ms, construct_again_ms, flatten_ms, create_ms, create_host_ms, create_device_ms, run_ms, median(t_prep), raw_ms, std::max(0.0, run_ms - raw_ms), median(t_build), median(t_destroy),
                median(t_op), median(t_old), sweeps0, static_cast<unsigned long long>(sum_view), map_equals_view ? "true" : "false",
                last ? "" : ",");
}

} // namespace

int main(int argc, char** argv)
{
    std::string configs = "alarm,dag,grid", dsc = "tests/golden/alarm_shaped.dsc";
    int reps = 0, grid_rows = 316;
    for(int i = 1; i < argc; ++i)
    {
        if(!std::strcmp(argv[i], "--configs") && i + 1 < argc) configs = argv[++i];
        else if(!std::strcmp(argv[i], "--dsc") && i + 1 < argc) dsc = argv[++i];
        else if(!std::strcmp(argv[i], "--reps") && i + 1 < argc) reps = std::atoi(argv[++i]);
        else if(!std::strcmp(argv[i], "--grid") && i + 1 < argc) grid_rows = std::atoi(argv[++i]);
        else if(!std::strcmp(argv[i], "--checksum")) checksum_only = true;
        else { std::fprintf(stderr, "usage: bench_dropin [--configs alarm,dag,grid] [--dsc FILE] [--reps N] [--grid ROWS]\n"); return 2; }
    }
    std::vector<std::string> todo;
    for(std::size_t a = 0; a <= configs.size();)
    {
        std::size_t const b = std::min(configs.find(',', a), configs.size());
        if(b > a) todo.push_back(configs.substr(a, b - a));
        a = b + 1;
    }
    try
    {
        std::printf("{");
        for(std::size_t c = 0; c < todo.size(); ++c)
        {
            bool const last = c + 1 == todo.size();
            double const t0 = now_ms();
            if(todo[c] == "alarm")
            {   // BASELINE configs[0]: the ALARM-shaped network through the DSC loader; 10 % evidence, eps 1e-6 (bench.py leg_alarm)
                bn::graph_t const g = bn::serializer::dsc().from_file(dsc);
                bench_network("config1_alarm", g, now_ms() - t0, 0.1, 1e-6, reps ? reps : 400, last);
            }
            else if(todo[c] == "dag")
            {   // BASELINE configs[1]: synth.random_dag(10000, 4, 64, 4, seed=1), 1 % evidence, eps 1e-3
                bn::graph_t const g = build_graph(dag_structure(10000, 4, 64, 4, 1), 1);
                bench_network("config2_dag", g, now_ms() - t0, 0.01, 1e-3, reps ? reps : 60, last);
            }
            else if(todo[c] == "grid")
            {   // BASELINE configs[2]: synth.grid(316, 316, 4, seed=2), 1 % evidence, eps 1e-3
                bn::graph_t const g = build_graph(grid_structure(grid_rows, grid_rows, 4), 2);
                bench_network("config3_grid", g, now_ms() - t0, 0.01, 1e-3, reps ? reps : 40, last);
            }
            else { std::fprintf(stderr, "unknown config %s\n", todo[c].c_str()); return 2; }
        }
        std::printf("}\n");
    }
    catch(std::exception const& ex)
    {
        std::printf("\n");
        std::fprintf(stderr, "bench_dropin: %s\n", ex.what());
        return 1;
    }
    return 0;
}

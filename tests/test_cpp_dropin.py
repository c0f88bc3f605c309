"""The header-only C++14 drop-in (include/bayesian/inference/*.hpp) over the C ABI.

CPU: the test program compiles against this repository's stand-in data model AND against the
reference's own graph.hpp / matrix.hpp (where /root/reference exists), and the flattened networks
equal the flat models used everywhere else.  GPU: the reference's seven BP test cases run through
bn::inference::belief_propagation, teacher tolerances of the reference tests, and the marginals are
bit-identical to the reference's golden outputs."""
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp")
LIBDIR = os.path.join(ROOT, "bayesiannetwork_amd")


def build(tmp_path, model_include, name):
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", model_include, SRC,
           "-L", LIBDIR, "-lbn_mi355x", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    return exe


def check_flatten(exe):
    from bayesiannetwork_amd import synth
    out = subprocess.run([exe, "--flatten"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout)
    for name, model in (("pearl", synth.pearl()), ("resume_chain", synth.resume_chain())):
        f = d[name]
        assert f["k"] == model.k.tolist() and f["in_ptr"] == model.in_ptr.tolist()
        assert f["in_idx"] == model.in_idx.tolist() and f["cpt_off"] == model.cpt_off.tolist()
        assert np.array_equal(np.asarray(f["cpt"]), model.cpt)


def test_flatten_with_compat_model(bnlib, tmp_path):
    check_flatten(build(tmp_path, os.path.join(ROOT, "include", "compat"), "dropin_compat"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/bayesian"), reason="reference headers not on this box")
def test_flatten_with_reference_model(bnlib, tmp_path):
    """True drop-in: the user's include path is the reference's; only bayesian/inference is ours."""
    check_flatten(build(tmp_path, "/root/reference", "dropin_ref"))


def check_dsc_flatten(exe):
    ref, _, _ = load_golden("bp_alarm_shaped")
    out = subprocess.run([exe, "--dsc", os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"), "--flatten"],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    f = json.loads(out.stdout)["net"]
    assert f["k"] == ref.k.tolist() and f["in_ptr"] == ref.in_ptr.tolist() and f["in_idx"] == ref.in_idx.tolist()
    assert np.array_equal(np.asarray(f["cpt"]), ref.cpt)


def test_dsc_loader_compat_equals_reference_golden(bnlib, tmp_path):
    """include/compat's serializer::dsc reads the ALARM-shaped file exactly as the reference's loader did."""
    check_dsc_flatten(build(tmp_path, os.path.join(ROOT, "include", "compat"), "dropin_compat_dsc"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/bayesian"), reason="reference headers not on this box")
def test_dsc_loader_reference_with_dropin_headers(bnlib, tmp_path):
    check_dsc_flatten(build(tmp_path, "/root/reference", "dropin_ref_dsc"))


@pytest.mark.gpu
def test_alarm_dsc_through_cpp_classes(bnlib, tmp_path):
    """BASELINE configs[0] end to end in C++: DSC file -> graph_t -> bn::inference::belief_propagation."""
    exe = build(tmp_path, os.path.join(ROOT, "include", "compat"), "dropin_dsc_gpu")
    out = subprocess.run([exe, "--dsc", os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(out.stdout)
    _, runs, _ = load_golden("bp_alarm_shaped")
    assert d["sweeps"] == runs[0]["sweeps"]
    assert np.abs(np.asarray(d["beliefs"]) - runs[0]["beliefs"]).max() < 1e-12


@pytest.mark.gpu
def test_reference_cases_through_cpp_classes(bnlib, tmp_path):
    exe = build(tmp_path, os.path.join(ROOT, "include", "compat"), "dropin_gpu")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    d = json.loads(out.stdout.splitlines()[0])
    _, pearl_runs, _ = load_golden("bp_pearl")
    assert np.array_equal(np.asarray(d["pearl_part1"]), pearl_runs[0]["beliefs"])
    assert np.array_equal(np.asarray(d["pearl_part2"]), pearl_runs[1]["beliefs"])
    assert d["pearl_part2_sweeps"] == pearl_runs[1]["sweeps"]
    _, chain_runs, _ = load_golden("bp_resume_chain")
    for i, r in enumerate(chain_runs):
        assert np.array_equal(np.asarray(d[f"resume_{i}"]), r["beliefs"])
    # CPT liveness (belief_propagation.hpp:61,186,252: the reference reads node->cpt per call): without reload() the functor
    # answers for the tables it was built from; with it, for the edited ones -- the oracle's bits on the edited network
    import oracle
    from bayesiannetwork_amd import Evidence, synth
    chain = synth.resume_chain()
    ev = Evidence.from_dict(chain, {3: 2})
    old = oracle.bp_run(chain, ev, 0.001)["beliefs"]
    cpt = chain.cpt.copy()
    cpt[chain.cpt_off[2] + 2:chain.cpt_off[2] + 4] = [0.1, 0.9]      # P(C | B = 1)
    edited = type(chain)(chain.k, chain.in_ptr, chain.in_idx, chain.cpt_off, cpt, name="resume_chain_edited")
    new = oracle.bp_run(edited, ev, 0.001)["beliefs"]
    assert np.array_equal(np.asarray(d["reload_before"]), old) and np.array_equal(np.asarray(d["reload_stale"]), old)
    assert np.array_equal(np.asarray(d["reload_fresh"]), new) and np.array_equal(np.asarray(d["reload_rebuilt"]), new)
    assert np.abs(new - old).max() > 0.05
    # bn::sampler::make_cpt: the fitted rows are exactly count / row total of the loaded table
    from bayesiannetwork_amd import synth
    pearl = synth.pearl()
    cnt, fit = np.asarray(d["fit_counts"]), np.asarray(d["fitted_cpt"])
    for v in range(pearl.n):
        c = cnt[pearl.cpt_off[v]:pearl.cpt_off[v + 1]].reshape(-1, pearl.k[v])
        want = np.where(c.sum(axis=1, keepdims=True) == 0, 1.0 / pearl.k[v], c / np.maximum(c.sum(axis=1, keepdims=True), 1))
        assert np.array_equal(fit[pearl.cpt_off[v]:pearl.cpt_off[v + 1]].reshape(-1, pearl.k[v]), want)
    # likelihood_weighting::make_samples through the C++ drop-in (seed 7, Pearl, H = 0, unit 200000, eps 0.005)
    # stops after the same number of units as the oracle's restatement of the reference loop
    # (likelihood_weighting.hpp:62-117) fed with the GPU's own stream; same pattern table, same marginals
    import oracle
    ev = np.array([-1, -1, -1, 0], np.int32)
    want = oracle.make_samples(pearl, ev, 200000, 0.005, seed=7, stream="repo")
    assert d["make_samples_units"] == want["units"] and d["make_samples_total"] == want["units"] * 200000
    tab = np.asarray(d["make_samples_table"], dtype=np.int64)
    tab = tab[np.lexsort(tab[:, :4].T[::-1])]
    assert np.array_equal(tab[:, :4], want["patterns"]) and np.array_equal(tab[:, 4], want["counts"].astype(np.int64))
    assert np.allclose(np.asarray(d["make_samples_marginals"]), want["marginals"], rtol=1e-9, atol=1e-15)
    # rejection_sampling through the C++ drop-in (seed 99): accepted counts and draws equal the oracle's
    from bayesiannetwork_amd import from_parent_lists
    net = from_parent_lists([2] * 5, [[], [0], [0], [1], [1, 2]],
                            [[.5, .5], [.8, .2, .1, .9], [.7, .3, .4, .6], [.6, .4, .1, .9], [.1, .9, .2, .8, .3, .7, .4, .6]])
    counts, drawn, acc = oracle.rs_run(net, np.array([0, -1, -1, 1, -1], np.int32), 10000, seed=99)
    assert d["rejection_drawn"] == drawn and acc == 10000
    assert np.array_equal(np.asarray(d["rejection"]), counts / 10000.0)


# ---- tests/cpp/bench_dropin.cpp: the C++ bench of the class surface (bench.py's `dropin_cpp` leg) ----

def _wsum64(arr, as_int=False):
    a = np.ascontiguousarray(arr)
    w = a.astype(np.int64).view(np.uint64) if as_int else a.astype(np.float64).view(np.uint64)
    with np.errstate(over="ignore"):
        return int((w * (np.arange(w.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64))


def _bench_networks():
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.dsc import load_dsc
    return {"config1_alarm": (load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))[0], 0.1, 1e-6),
            "config2_dag": (synth.random_dag(10000, 4, 64, 4, seed=1), 0.01, 1e-3),
            "config3_grid": (synth.grid(316, 316, 4, seed=2), 0.01, 1e-3)}


def test_bench_dropin_builds_the_bench_networks(bnlib):
    """The C++ bench builds BASELINE configs[0..2] through graph_t / cpt_t with the generators of synth.py restated in C++:
    flat model and evidence of query 0 equal the Python side's (checksums over every parent index, CPT entry, evidence pair)."""
    import __graft_entry__
    from bayesiannetwork_amd import synth
    exe = __graft_entry__.build_bench_dropin()
    out = subprocess.run([exe, "--checksum"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout)
    for key, (g, frac, _) in _bench_networks().items():
        st = synth.random_evidence(g, frac, seed=7).hard_states(g)
        nodes = np.nonzero(st >= 0)[0]
        assert d[key]["nodes"] == g.n and d[key]["edges"] == g.n_edges
        assert d[key]["in_idx"] == f"{_wsum64(g.in_idx, True):016x}"
        assert d[key]["cpt"] == f"{_wsum64(g.cpt):016x}"
        assert d[key]["evidence"] == f"{_wsum64(nodes * 256 + st[nodes], True):016x}"


@pytest.mark.gpu
def test_bench_dropin_view_equals_map_equals_c_abi(bnlib):
    """run()'s marginals_view == operator()'s map bit for bit on configs[0..2] at full size, and both equal what the ctypes
    path gets for the same network and evidence (sweep count, checksum over all marginals)."""
    import __graft_entry__
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import Engine
    exe = __graft_entry__.build_bench_dropin()
    out = subprocess.run([exe, "--reps", "3", "--dsc", os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc")],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    for key, (g, frac, eps) in _bench_networks().items():
        assert d[key]["map_equals_view"] is True
        with Engine(g, device=0) as eng:
            r = eng.bp_run(synth.random_evidence(g, frac, seed=7), eps)   # (a copy: bp_run_view's array dies with the engine)
        assert r["sweeps"] == d[key]["sweeps_query0"]
        assert f"{_wsum64(r['beliefs']):016x}" == d[key]["wsum64_query0"]
        assert d[key]["run_view_ms"] > 0 and d[key]["operator_ms"] >= d[key]["run_view_ms"]

"""Python bindings for the CPU checker -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
package.  ``bp_run`` / ``lw_run`` call the plain-C restatement in ``liboracle.so``
(``bp_oracle.c`` / ``lw_oracle.c``); ``ref_bp`` / ``ref_lw`` execute ``_ref/ref_driver``,
the unmodified reference headers compiled where they lie (exists only in the container
that has ``/root/reference``).
"""
from __future__ import annotations

import ctypes
import json
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
REF_DRIVER = os.path.join(_HERE, "_ref", "ref_driver")


def build(quiet: bool = True) -> None:
    """make liboracle.so (and _ref/ref_driver when /root/reference exists)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        path = os.environ.get("BN_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # BN_ORACLE_LIB: the sanitizer build (scripts/san_cpu.sh)
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        i32p, i64p, f64p = (ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64),
                            ctypes.POINTER(ctypes.c_double))
        L.oracle_bp_run_mt.restype = ctypes.c_int
        L.oracle_bp_run_mt.argtypes = [ctypes.c_int, i32p, i32p, i32p, i64p, f64p, ctypes.c_int, i32p, i32p,
                                       f64p, ctypes.c_double, ctypes.c_int, f64p, ctypes.POINTER(ctypes.c_int),
                                       f64p, ctypes.c_int, f64p, ctypes.c_int]
        L.oracle_lw_run.restype = ctypes.c_int
        L.oracle_lw_run.argtypes = [ctypes.c_int, i32p, i32p, i32p, i64p, f64p, i32p, i32p, ctypes.c_uint64,
                                    ctypes.c_uint64, ctypes.c_uint64, f64p, ctypes.POINTER(ctypes.c_uint8),
                                    f64p, ctypes.c_uint64]
        L.oracle_philox4x32_10.restype = None
        _LIB = L
    return _LIB


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def bp_run(model, evidence=None, eps: float = 0.001, max_sweeps: int = 0, threads: int = 1,
           dump_msgs: bool = False, res_cap: int = 4096):
    """Restated reference BP.  Returns dict(beliefs, sweeps, residuals[, pi_msg, lambda_msg])."""
    from bayesiannetwork_amd.flat import Evidence
    ev = evidence if evidence is not None else Evidence.none()
    L = lib()
    bel = np.zeros(int(model.k.sum()), dtype=np.float64)
    res = np.zeros(res_cap, dtype=np.float64)
    sweeps = ctypes.c_int(0)
    nm = int(model.k[model.in_idx].sum()) if model.n_edges else 0
    dump = np.zeros(2 * nm if dump_msgs else 1, dtype=np.float64)
    rc = L.oracle_bp_run_mt(model.n, _p(model.k, ctypes.c_int32), _p(model.in_ptr, ctypes.c_int32),
                            _p(model.in_idx, ctypes.c_int32), _p(model.cpt_off, ctypes.c_int64),
                            _p(model.cpt, ctypes.c_double), ev.ne, _p(ev.node, ctypes.c_int32),
                            _p(ev.off, ctypes.c_int32), _p(ev.val, ctypes.c_double), float(eps),
                            int(max_sweeps), _p(bel, ctypes.c_double), ctypes.byref(sweeps),
                            _p(res, ctypes.c_double), res_cap,
                            _p(dump, ctypes.c_double) if dump_msgs else None, int(threads))
    if rc != 0:
        raise RuntimeError(f"oracle_bp_run failed: {rc}")
    out = {"beliefs": bel, "sweeps": sweeps.value, "residuals": res[:min(sweeps.value, res_cap)].copy()}
    if dump_msgs:
        out["pi_msg"], out["lambda_msg"] = dump[:nm].copy(), dump[nm:].copy()
    return out


def lw_run(model, ev_state, n_samples: int, seed: int, s_begin: int = 0, topo=None, states_cap: int = 0):
    """Restated LW with the repository's random stream (Philox-seeded xoshiro128++ per sample).  Returns dict(hist[, states, weights])."""
    L = lib()
    n = model.n
    topo = np.arange(n, dtype=np.int32) if topo is None else np.ascontiguousarray(topo, dtype=np.int32)
    ev_state = np.ascontiguousarray(ev_state, dtype=np.int32)
    hist = np.zeros(int(model.k.sum()), dtype=np.float64)
    states = np.zeros(max(1, states_cap * n), dtype=np.uint8)
    weights = np.zeros(max(1, states_cap), dtype=np.float64)
    rc = L.oracle_lw_run(n, _p(model.k, ctypes.c_int32), _p(model.in_ptr, ctypes.c_int32),
                         _p(model.in_idx, ctypes.c_int32), _p(model.cpt_off, ctypes.c_int64),
                         _p(model.cpt, ctypes.c_double), _p(topo, ctypes.c_int32), _p(ev_state, ctypes.c_int32),
                         ctypes.c_uint64(s_begin), ctypes.c_uint64(n_samples), ctypes.c_uint64(seed),
                         _p(hist, ctypes.c_double), _p(states, ctypes.c_uint8), _p(weights, ctypes.c_double),
                         ctypes.c_uint64(states_cap))
    if rc != 0:
        raise RuntimeError(f"oracle_lw_run failed: {rc}")
    out = {"hist": hist}
    if states_cap:
        out["states"] = states[:states_cap * n].reshape(states_cap, n)
        out["weights"] = weights[:states_cap]
    return out


def rs_run(model, ev_state, n_accept: int, seed: int, max_draw: int = 1 << 34, s_begin: int = 0, topo=None):
    """Restated rejection sampling (oracle_rs_run).  Returns (counts, drawn, accepted)."""
    L = lib()
    L.oracle_rs_run.restype = ctypes.c_int
    n = model.n
    topo = np.arange(n, dtype=np.int32) if topo is None else np.ascontiguousarray(topo, dtype=np.int32)
    ev_state = np.ascontiguousarray(ev_state, dtype=np.int32)
    counts = np.zeros(int(model.k.sum()), dtype=np.float64)
    drawn, acc = ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = L.oracle_rs_run(n, _p(model.k, ctypes.c_int32), _p(model.in_ptr, ctypes.c_int32),
                         _p(model.in_idx, ctypes.c_int32), _p(model.cpt_off, ctypes.c_int64),
                         _p(model.cpt, ctypes.c_double), _p(topo, ctypes.c_int32), _p(ev_state, ctypes.c_int32),
                         ctypes.c_uint64(s_begin), ctypes.c_uint64(n_accept), ctypes.c_uint64(max_draw),
                         ctypes.c_uint64(seed), _p(counts, ctypes.c_double), ctypes.byref(drawn), ctypes.byref(acc))
    if rc != 0:
        raise RuntimeError(f"oracle_rs_run failed: {rc}")
    return counts, drawn.value, acc.value


def make_cpt(model, patterns, counts):
    """Restated sampler::make_cpt (oracle_make_cpt; parity unpinned: the reference file needs Boost)."""
    L = lib()
    L.oracle_make_cpt.restype = ctypes.c_int
    patterns = np.ascontiguousarray(patterns, dtype=np.uint8).reshape(-1, model.n)
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    out = np.zeros(int(model.cpt_off[-1]), dtype=np.float64)
    rc = L.oracle_make_cpt(model.n, _p(model.k, ctypes.c_int32), _p(model.in_ptr, ctypes.c_int32),
                           _p(model.in_idx, ctypes.c_int32), _p(model.cpt_off, ctypes.c_int64),
                           ctypes.c_int64(patterns.shape[0]), _p(patterns, ctypes.c_uint8),
                           _p(counts, ctypes.c_uint64), _p(out, ctypes.c_double))
    if rc != 0:
        raise RuntimeError(f"oracle_make_cpt failed: {rc}")
    return out


def lw_normalize(model, hist):
    """likelihood_weighting.hpp:197-221 applied per node."""
    out = hist.astype(np.float64).copy()
    off = model.node_off
    for v in range(model.n):
        h = out[off[v]:off[v + 1]]
        s = 0.0
        for x in h:
            s += x
        if s < 1.0e-20:
            h[:] = 1.0 / h.size
        else:
            h /= s
    return out


def philox(ctr, key):
    c = (ctypes.c_uint32 * 4)(*ctr)
    k = (ctypes.c_uint32 * 2)(*key)
    o = (ctypes.c_uint32 * 4)()
    lib().oracle_philox4x32_10(c, k, o)
    return list(o)


def xoshiro128pp(state, n: int):
    """n 32-bit outputs of xoshiro128++ from `state` (4 words); returns (outputs, final state)."""
    L = lib()
    L.oracle_xoshiro128pp_next.restype = ctypes.c_uint32
    x = (ctypes.c_uint32 * 4)(*state)
    out = [int(L.oracle_xoshiro128pp_next(x)) for _ in range(n)]
    return out, list(x)


def lw_uniform(seed: int, sample: int, position: int) -> float:
    """u(sample, position) of the sampler's stream (see lw_oracle.c)."""
    L = lib()
    L.oracle_lw_uniform.restype = ctypes.c_double
    return float(L.oracle_lw_uniform(ctypes.c_uint64(seed), ctypes.c_uint64(sample), ctypes.c_uint32(position)))


# ---- replay of the reference's sampler family (ref_replay.c) -------------------------

def _model_args(model):
    return (model.n, _p(model.k, ctypes.c_int32), _p(model.in_ptr, ctypes.c_int32), _p(model.in_idx, ctypes.c_int32),
            _p(model.cpt_off, ctypes.c_int64), _p(model.cpt, ctypes.c_double))


def mt19937_words(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.uint32)
    lib().oracle_mt19937_words(ctypes.c_uint32(seed), n, _p(out, ctypes.c_uint32))
    return out


def mt19937_uniforms(seed: int, n: int) -> np.ndarray:
    """std::uniform_real_distribution<double>(0,1) on std::mt19937(seed), libstdc++."""
    out = np.zeros(n, dtype=np.float64)
    lib().oracle_mt19937_uniforms(ctypes.c_uint32(seed), n, _p(out, ctypes.c_double))
    return out


def ref_visit_order(model) -> np.ndarray:
    """Order in which weighted_sample / generate_pattern sample the vertices (likelihood_weighting.hpp:162-170)."""
    out = np.zeros(max(model.n, 1), dtype=np.int32)
    rc = lib().oracle_ref_visit_order(model.n, _p(model.in_ptr, ctypes.c_int32), _p(model.in_idx, ctypes.c_int32),
                                      _p(out, ctypes.c_int32))
    if rc != 0:
        raise RuntimeError(f"oracle_ref_visit_order failed: {rc}")
    return out[:model.n]


def ref_lw_replay(model, ev_state, n_samples: int, mt_seed: int) -> np.ndarray:
    """likelihood_weighting::operator() replayed with mt19937(mt_seed): the reference's marginals, bit for bit."""
    ev_state = np.ascontiguousarray(ev_state, dtype=np.int32)
    out = np.zeros(int(model.k.sum()), dtype=np.float64)
    L = lib()
    L.oracle_ref_lw_run.restype = ctypes.c_int
    rc = L.oracle_ref_lw_run(*_model_args(model), _p(ev_state, ctypes.c_int32), ctypes.c_uint64(n_samples),
                             ctypes.c_uint32(mt_seed), _p(out, ctypes.c_double))
    if rc != 0:
        raise RuntimeError(f"oracle_ref_lw_run failed: {rc}")
    return out


def make_samples(model, ev_state, unit_size: int, eps: float, seed: int, stream: str = "mt19937",
                 order=None, sample_begin: int = 0, max_units: int = 0, pat_cap: int = 1 << 20):
    """likelihood_weighting::make_samples (:62-117).  stream "mt19937": replay of the reference with its
    engine reseeded mt19937(seed) and its own visiting order; stream "repo": the same loop fed with this
    repository's per-sample streams in visiting order `order` (default: identity = the GPU's order for
    models whose parents precede their children).  Returns dict(units, marginals, patterns [P][n], counts [P])."""
    ev_state = np.ascontiguousarray(ev_state, dtype=np.int32)
    kind = {"mt19937": 0, "repo": 1}[stream]
    if kind == 1 and order is None:
        order = np.arange(model.n, dtype=np.int32)
    ordp = None if order is None else _p(np.ascontiguousarray(order, dtype=np.int32), ctypes.c_int32)
    marg = np.zeros(int(model.k.sum()), dtype=np.float64)
    pats = np.zeros((pat_cap, max(model.n, 1)), dtype=np.uint8)
    cnts = np.zeros(pat_cap, dtype=np.uint64)
    units, npat = ctypes.c_uint64(0), ctypes.c_uint64(0)
    L = lib()
    L.oracle_make_samples.restype = ctypes.c_int
    rc = L.oracle_make_samples(*_model_args(model), ordp, _p(ev_state, ctypes.c_int32), ctypes.c_uint64(unit_size),
                               ctypes.c_double(eps), ctypes.c_uint64(max_units), kind, ctypes.c_uint64(seed),
                               ctypes.c_uint64(sample_begin), ctypes.byref(units), _p(marg, ctypes.c_double),
                               ctypes.c_uint64(pat_cap), _p(pats, ctypes.c_uint8), _p(cnts, ctypes.c_uint64),
                               ctypes.byref(npat))
    if rc not in (0, 1):
        raise RuntimeError(f"oracle_make_samples failed: {rc}")
    if npat.value > pat_cap:
        raise RuntimeError("pattern table truncated: raise pat_cap")
    return {"units": units.value, "marginals": marg, "patterns": pats[:npat.value, :model.n].copy(),
            "counts": cnts[:npat.value].copy(), "hit_max_units": rc == 1}


def ref_rs_replay(model, cond_state, num: int, mt_seed: int, max_draw: int = 0):
    """rejection_sampling::operator() replayed with mt19937(mt_seed).  Returns (marginals, drawn)."""
    cond_state = np.ascontiguousarray(cond_state, dtype=np.int32)
    out = np.zeros(int(model.k.sum()), dtype=np.float64)
    drawn = ctypes.c_uint64(0)
    L = lib()
    L.oracle_ref_rs_run.restype = ctypes.c_int
    rc = L.oracle_ref_rs_run(*_model_args(model), _p(cond_state, ctypes.c_int32), ctypes.c_uint64(num),
                             ctypes.c_uint32(mt_seed), ctypes.c_uint64(max_draw), _p(out, ctypes.c_double),
                             ctypes.byref(drawn))
    if rc == 2:
        raise IndexError("choice_pattern: uniform not below the row total (the reference throws std::out_of_range)")
    if rc not in (0, 1):
        raise RuntimeError(f"oracle_ref_rs_run failed: {rc}")
    return out, drawn.value


# ---- the real reference (only where /root/reference exists) -------------------------

def ref_available() -> bool:
    return os.path.exists(REF_DRIVER)


def _run_ref(text: str, timeout: float):
    with tempfile.NamedTemporaryFile("w", suffix=".bnflat", delete=False) as f:
        f.write(text)
        path = f.name
    try:
        p = subprocess.run([REF_DRIVER, path], capture_output=True, text=True, timeout=timeout)
    finally:
        os.unlink(path)
    if p.returncode != 0:
        raise RuntimeError(f"ref_driver exit {p.returncode}: {p.stderr[-500:]}")
    return json.loads(p.stdout)


def ref_bp(model, evidence=None, eps: float = 0.001, dump_msgs: bool = False, timeout: float = 3600):
    from bayesiannetwork_amd.flat import Evidence
    ev = evidence if evidence is not None else Evidence.none()
    text = model.to_bnflat_text() + f"bp {eps!r} {1 if dump_msgs else 0}\n" + ev.to_bnflat_text()
    return _run_ref(text, timeout)


def ref_dsc_bp(dsc_path: str, evidence, eps: float, dump_msgs: bool = False, timeout: float = 600):
    """The reference's own DSC loader (serializer/dsc.hpp) + BP on a .dsc file."""
    with tempfile.NamedTemporaryFile("w", suffix=".req", delete=False) as f:
        f.write(f"bp {eps!r} {1 if dump_msgs else 0}\n" + evidence.to_bnflat_text())
        req = f.name
    try:
        p = subprocess.run([REF_DRIVER, "--dsc", dsc_path, req], capture_output=True, text=True, timeout=timeout)
    finally:
        os.unlink(req)
    if p.returncode != 0:
        raise RuntimeError(f"ref_driver exit {p.returncode}: {p.stderr[-500:]}")
    return json.loads(p.stdout)


def ref_lw(model, ev_state, n_samples: int, seed: int, timeout: float = 3600):
    pairs = [(v, int(s)) for v, s in enumerate(ev_state) if s >= 0]
    text = model.to_bnflat_text() + f"lw {n_samples} {seed}\n{len(pairs)}\n" + \
        "".join(f"{v} {s}\n" for v, s in pairs)
    return _run_ref(text, timeout)


def _pairs_text(ev_state):
    pairs = [(v, int(s)) for v, s in enumerate(ev_state) if s >= 0]
    return f"{len(pairs)}\n" + "".join(f"{v} {s}\n" for v, s in pairs)


def ref_make_samples(model, ev_state, unit_size: int, eps: float, seed: int, timeout: float = 3600):
    """The reference's own likelihood_weighting::make_samples, engine reseeded mt19937(seed)."""
    text = model.to_bnflat_text() + f"lwms {unit_size} {eps!r} {seed}\n" + _pairs_text(ev_state)
    r = _run_ref(text, timeout)
    tab = np.asarray(r["patterns"], dtype=np.int64).reshape(-1, model.n + 1)
    order = np.lexsort(tab[:, :model.n].T[::-1])
    r["patterns"] = tab[order, :model.n].astype(np.uint8)
    r["counts"] = tab[order, model.n].astype(np.uint64)
    return r


def ref_rs(model, cond_state, num: int, seed: int, timeout: float = 3600):
    """The reference's own rejection_sampling::operator(), engine reseeded mt19937(seed)."""
    text = model.to_bnflat_text() + f"rs {num} {seed}\n" + _pairs_text(cond_state)
    return _run_ref(text, timeout)

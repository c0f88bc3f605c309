"""Shared helpers for the parity tests: golden fixture loading and comparison."""
import glob
import os

import numpy as np

from bayesiannetwork_amd import Evidence, FlatModel

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(prefix="bp_"):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = FlatModel(z["k"], z["in_ptr"], z["in_idx"], z["cpt_off"], z["cpt"], name=name)
    runs = []
    for i in range(int(z["n_runs"])):
        pre = f"run{i}_"
        r = {key[len(pre):]: z[key] for key in z.files if key.startswith(pre)}
        r["evidence"] = Evidence(r["ev_node"], r["ev_off"], r["ev_val"])
        r["eps"] = float(r["eps"])
        r["sweeps"] = int(r["sweeps"])
        runs.append(r)
    extra = {key: z[key] for key in z.files if not key.startswith("run") and key not in
             ("k", "in_ptr", "in_idx", "cpt_off", "cpt", "n_runs")}
    return model, runs, extra


def max_parents(model):
    return int(np.diff(model.in_ptr).max()) if model.n else 0


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny) over entries; NaNs must coincide."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    nan = np.isnan(b)
    assert (np.isnan(a) == nan).all()
    if nan.all():
        return 0.0
    d = np.abs(a[~nan] - b[~nan]) / np.maximum(np.abs(b[~nan]), 1e-300)
    d[(a[~nan] == b[~nan])] = 0.0
    return float(d.max()) if d.size else 0.0


def margin_ok(residuals, eps, ulp=1e-13):
    """True when no per-sweep residual sits within rounding distance of eps, i.e. the stopping
    sweep cannot flip on last-bit differences ('stopping at the same sweep')."""
    r = np.asarray(residuals, float)
    return bool((np.abs(r - eps) > ulp * max(eps, 1e-300)).all())


def hub_network(n_children, seed=77):
    """k = 4: one node with `n_children` children, each of which has one more parent of its own (a root)"""
    from bayesiannetwork_amd import from_parent_lists
    from bayesiannetwork_amd.synth import uniform01
    parents = [[]] + [[] for _ in range(n_children)] + [[0, 1 + c] for c in range(n_children)]
    cpts, at = [], 0
    for ps in parents:
        rows = 4 ** len(ps)
        r = 0.1 + 0.9 * uniform01(seed, at, rows * 4).reshape(rows, 4)
        at += rows * 4
        cpts.append((r / r.sum(axis=1, keepdims=True)).ravel().tolist())
    return from_parent_lists(k=[4] * len(parents), parents=parents, cpts=cpts, name=f"hub{n_children}")


def parse_bench_output(stdout):
    """bench.py's stdout -> (contract line as a dict, {extra name: record}).  The contract line is the LAST line of the
    output (what the driver parses); the extras are the `{"extra": ..., "record": ...}` lines before it."""
    import json
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    line = json.loads(lines[-1])
    assert "metric" in line, "the last stdout line is not the contract line"
    extras = {}
    for ln in lines[:-1]:
        if ln.startswith('{"extra"'):
            d = json.loads(ln)
            extras[d["extra"]] = d["record"]
    return line, extras

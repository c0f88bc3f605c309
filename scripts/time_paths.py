#!/usr/bin/env python3
"""Every execution path an engine is eligible for, forced one after another on the same staged query, against the path the engine
picks by default and the one option "autotune" keeps: us per query (host wall clock, evidence staged) and us per sweep (device
clock).  GPU box only:  python scripts/time_paths.py [--json gpurun_out/r05_paths.json] [names...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import synth  # noqa: E402
from bayesiannetwork_amd.dsc import load_dsc  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

FORCE = {0: {"multisweep": 0}, 2: {"multisweep": 2, "small": 0, "mid": 0, "dag": 0}, 3: {"small": 2, "mid": 0, "dag": 0},
         4: {"mid": 2, "small": 0, "dag": 0}, 5: {"dag": 2}}
DEFAULTS = {"multisweep": 1, "small": 1, "mid": 1, "dag": 1}
ELIGIBLE = {2: "resident_eligible", 3: "small_eligible", 4: "mid_eligible", 5: "dag_eligible"}


def networks():
    alarm, _ = load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))
    return [("alarm_shaped", alarm), ("pearl", synth.pearl()), ("grid8", synth.grid(8, 8, 4, seed=1)), ("grid16", synth.grid(16, 16, 4, seed=1)),
            ("grid32", synth.grid(32, 32, 4, seed=1)), ("grid40", synth.grid(40, 40, 4, seed=1)),
            ("dag200", synth.random_dag(200, 4, 64, 4, seed=200)), ("dag1000", synth.random_dag(1000, 4, 64, 4, seed=1000)),
            ("mixed60", synth.random_dag(60, 3, 16, [2, 3, 4, 3, 2, 4, 4], seed=9)),
            ("mixed300", synth.random_dag(300, 3, 32, [2, 3, 4, 3, 2, 5], seed=4)),
            ("mixed2k", synth.random_dag(2000, 4, 64, [2, 3, 4, 3, 2, 4, 4], seed=9)),
            ("chain200", synth.grid(200, 1, 4, seed=5)), ("grid64", synth.grid(64, 64, 4, seed=1)),
            ("grid128", synth.grid(128, 128, 4, seed=1)), ("grid200", synth.grid(200, 200, 4, seed=1)),
            ("grid250", synth.grid(250, 250, 4, seed=1)), ("grid316", synth.grid(316, 316, 4, seed=2)),
            ("dag2p_3000", synth.random_dag(3000, 2, 64, 4, seed=3)), ("dag3000", synth.random_dag(3000, 4, 64, 4, seed=5)),
            ("dag10k", synth.random_dag(10000, 4, 64, 4, seed=1))]


def measure(eng, eps, reps=30):
    for _ in range(4):
        eng.bp_run_device(eps)
    t0 = time.perf_counter()
    dev, sweeps = 0.0, 0
    for _ in range(reps):
        r = eng.bp_run_device(eps)
        dev += eng.bp_stats()["sweep_devclock_ms"]
        sweeps += r["sweeps"]
    return {"path": eng.last_path(), "us_per_query": (time.perf_counter() - t0) / reps * 1e6, "us_per_sweep": dev / sweeps * 1e3}


def one(name, g, eps=1e-6):
    ev = synth.random_evidence(g, 0.02, seed=7)
    row = {"nodes": g.n}
    with Engine(g) as eng:
        eng.bp_set_evidence(ev)
        row["default"] = measure(eng, eps)
        paths = {}
        for path, opts in FORCE.items():
            if path in ELIGIBLE and not eng.info(ELIGIBLE[path]):
                continue
            for k, v in {**DEFAULTS, **opts}.items():
                eng.set_option(k, v)
            m = measure(eng, eps)
            if m["path"] == path:
                paths[path] = m
        for k, v in DEFAULTS.items():
            eng.set_option(k, v)
        eng.set_option("autotune", 1)
        row["autotuned"] = measure(eng, eps)
        row["autotuned"]["chosen"] = eng.info("autotuned_path")
        row["paths"] = {str(k): v for k, v in paths.items()}
    best = min(paths.values(), key=lambda m: m["us_per_query"])
    row["best_path"] = best["path"]
    row["default_over_best"] = row["default"]["us_per_query"] / best["us_per_query"]
    row["autotuned_over_best"] = row["autotuned"]["us_per_query"] / best["us_per_query"]
    return row


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_path = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    if out_path in args:
        args.remove(out_path)
    out = {}
    for name, g in networks():
        if args and name not in args:
            continue
        out[name] = one(name, g)
        r = out[name]
        print(f"{name:14s} default path {r['default']['path']} {r['default']['us_per_query']:7.1f} us | autotuned path {r['autotuned']['chosen']} "
              f"{r['autotuned']['us_per_query']:7.1f} us | " + "  ".join(f"{k}: {v['us_per_query']:.1f}" for k, v in r["paths"].items())
              + f" | default/best {r['default_over_best']:.2f} autotuned/best {r['autotuned_over_best']:.2f}", flush=True)
    if out_path:
        import hashlib
        from bayesiannetwork_amd import _lib
        out["lib_sha256"] = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()   # the build these timings belong to (bench.py checks it)
        json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()

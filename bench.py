#!/usr/bin/env python3
"""bench.py -- edge-messages/sec to BP convergence on the BASELINE.json grid (config 3).

A "step" is one belief-propagation run to convergence (init, evidence, sweeps until
maximum_difference < eps, beliefs) on the 316x316 k=4 grid BN with 1 % hard evidence, model and
evidence already resident in HBM.  value = 2E * sweeps * steps / wall time, whole job.
One JSON line on stdout (rank 0).  See DESIGN.md "Measurement" for the definitions.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def profiled_traffic(rows, cols):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC passes (scripts/profile_bench.sh:
    FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950).  Counters cannot be collected from inside this process, so the number comes from
    the last profile of the same command; None when that workload was not profiled."""
    label = {(316, 316): "grid316", (2048, 2048): "grid2048"}.get((rows, cols))
    best = None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        if name.endswith("_summary.json") and label:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            if f"{label}_traffic_bytes_per_launch" in d:
                best = (d[f"{label}_traffic_bytes_per_launch"], f"profiles/{name}")
    return best


def cpu_baseline(model, ev, eps, budget_s=12.0, threads=1):
    """The oracle (plain-C port of the reference algorithm) timed on this box's host cores on a
    bounded number of full runs of the same workload: 1 thread like the reference, or OpenMP over
    the nodes of a sweep (both are reported)."""
    import oracle
    t0 = time.perf_counter()
    runs, msgs = 0, 0
    while True:
        r = oracle.bp_run(model, ev, eps, threads=threads)
        runs += 1
        msgs += model.messages_per_sweep() * r["sweeps"]
        if time.perf_counter() - t0 > budget_s or runs >= 8:
            break
    dt = time.perf_counter() - t0
    return {"value": msgs / dt, "unit": "edge-messages/s", "cores": threads, "kind": "port",
            "sample": f"{runs} full runs of the same workload ({r['sweeps']} sweeps each), oracle/bp_oracle.c, "
                      f"{threads} thread{'s' if threads > 1 else ''}"}


def bench_lw(a, local_rank, torch):
    """BASELINE configs[4] on one GPU: weighted samples/s on the 10 k-node DAG with 1 % evidence."""
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import Engine
    d = synth.random_dag(10000, 4, 64, 4, seed=1)
    ev = synth.random_evidence(d, a.evidence, seed=7).hard_states(d)
    eng = Engine(d, device=local_rank)
    for w in range(max(a.warmup, 1)):
        eng.lw_run(ev, a.samples, seed=1, sample_begin=w * a.samples)
    torch.cuda.synchronize()
    steps = max(1, min(a.steps, 10))
    t0 = time.perf_counter()
    for i in range(steps):
        eng.lw_run(ev, a.samples, seed=1, sample_begin=(i + a.warmup) * a.samples)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rate = a.samples * steps / dt
    # algorithmic traffic: per node-sample 1 B state written, its parents' states read, 1 B re-read
    # by the histogram pass (informational, CPTs are cache-resident)
    bytes_per_sample = d.n * 2 + d.n_edges
    out = {"metric": "weighted samples/sec (likelihood weighting)", "value": rate, "unit": "samples/s",
           "n_gpus": 1, "steps": steps, "warmup": a.warmup, "ms_per_step": dt / steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"likelihood weighting, 10 k-node random DAG, {int((ev >= 0).sum())} evidence nodes, "
                                  f"{a.samples} samples per step (BASELINE.json configs[4])",
                      "node_samples_per_s": rate * d.n},
           "roofline": {"bound": "hbm", "achieved": rate * bytes_per_sample / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": rate * bytes_per_sample / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "lw_sample_kernel + lw_hist_kernel",
                        "note": "informational: the sampler is VALU/latency-bound (DESIGN.md section 4), not HBM-bound"}}
    if not a.no_cpu:
        import oracle
        t0 = time.perf_counter()
        n_cpu = 2000
        oracle.lw_run(d, ev, n_cpu, seed=1)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n_cpu / dtc, "unit": "samples/s", "cores": 1, "kind": "port",
                               "sample": f"{n_cpu} samples of the same workload, oracle/lw_oracle.c, 1 thread"}
    print(json.dumps(out))


def cpu_reference_small():
    """The UNMODIFIED reference (oracle/_ref/ref_driver, built where /root/reference exists and
    shipped as a binary) on the largest grid it finishes in seconds: its dense V x V graph_t makes
    run time grow like V^3, so the bench workload itself (10^5 nodes, 160 GB of
    adjacency matrix) is out of its reach.  Informational, next to the port's number."""
    import oracle
    from bayesiannetwork_amd import synth
    if not oracle.ref_available():
        return None
    g = synth.grid(16, 16, 4, seed=116)
    try:
        r = oracle.ref_bp(g, None, 1e-3, timeout=120)
    except Exception as ex:  # noqa: BLE001 - informational leg only
        return {"error": str(ex)[:200]}
    return {"value": g.messages_per_sweep() * r["sweeps"] / r["sweep_s"], "unit": "edge-messages/s", "cores": 1,
            "kind": "reference", "sample": f"16x16 k=4 grid ({g.n} nodes), {r['sweeps']} sweeps, "
                                           "bn::inference::belief_propagation unmodified, 1 thread"}


def measured_stream_gbs(torch):
    """Achievable HBM rate on this box (report against the nominal peak AND a measured
    stream figure): device-to-device copy of 1 GiB (read + write), best of 5, outside the timed region."""
    try:
        n = 1 << 27  # doubles
        src = torch.ones(n, dtype=torch.float64, device="cuda")
        dst = torch.empty_like(src)
        best = 0.0
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dst.copy_(src)
            e1.record()
            e1.synchronize()
            best = max(best, 2 * n * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del src, dst
        return best
    except Exception:  # noqa: BLE001 - informational only
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["grid", "dag", "lw"], default="grid",
                    help="grid = BASELINE configs[2] (headline); dag = configs[1], 10 k-node random DAG; "
                         "lw = configs[4], likelihood weighting on the 10 k-node DAG")
    ap.add_argument("--samples", type=int, default=2000000, help="lw: weighted samples per step")
    ap.add_argument("--rows", type=int, default=316)
    ap.add_argument("--cols", type=int, default=316)
    ap.add_argument("--eps", type=float, default=1e-3)
    ap.add_argument("--evidence", type=float, default=0.01)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-weak", dest="no_weak", action="store_true", help="N>1: skip the weak-scaling extra")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import torch  # device plumbing only: barrier / synchronize around the timed region
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import Engine

    if world > 1 or os.environ.get("BN_FORCE_MULTI"):
        from bayesiannetwork_amd import multigpu
        if a.workload == "lw":
            return multigpu.bench_lw_main(a, rank, world, local_rank)
        return multigpu.bench_main(a, rank, world, local_rank)

    torch.cuda.set_device(local_rank)
    if a.workload == "lw":
        return bench_lw(a, local_rank, torch)
    if a.workload == "dag":
        g = synth.random_dag(10000, 4, 64, 4, seed=1)
        wname = (f"10 k-node random DAG, <=4 parents, k=4, {g.n_edges} edges (BASELINE.json configs[1])")
    else:
        g = synth.grid(a.rows, a.cols, 4, seed=2)
        wname = (f"{a.rows}x{a.cols} 2D-grid BN, k=4, {g.n} nodes, {g.n_edges} edges (BASELINE.json configs[2])")
    ev = synth.random_evidence(g, a.evidence, seed=7)
    eng = Engine(g, device=local_rank)
    eng.bp_set_evidence(ev)  # inputs resident in HBM before the timed region
    for _ in range(max(a.warmup, 1)):
        r = eng.bp_run_device(a.eps)
    torch.cuda.synchronize()
    sweeps_total, kern_ms, launches = 0, 0.0, 0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = eng.bp_run_device(a.eps)
        st = eng.bp_stats()
        sweeps_total += r["sweeps"]
        kern_ms += st["sweep_kernel_ms"]
        launches += st["sweep_launches"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.bp_stats()
    msgs = g.messages_per_sweep() * sweeps_total
    avg_launch_s = kern_ms * 1e-3 / max(launches, 1)
    if avg_launch_s <= 0:  # BN_TIMING=0: no HIP events; fall back to the whole-run clock (upper bound)
        avg_launch_s = dt / max(launches, 1)
    achieved = st["algorithmic_bytes_per_sweep"] / avg_launch_s / 1e9
    traffic = profiled_traffic(a.rows, a.cols) if a.workload == "grid" else None
    out = {
        "metric": "edge-messages/sec to BP convergence", "value": msgs / dt, "unit": "edge-messages/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{wname}, {ev.ne} evidence nodes, eps={a.eps:g}",
                   "sweeps_per_step": sweeps_total / a.steps, "messages_per_sweep": g.messages_per_sweep(),
                   "parallelism": "1 GPU"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0] if traffic else None,
                     "traffic_source": traffic[1] if traffic else None,
                     "kernel": "bp_sweep_kernel", "avg_launch_us": avg_launch_s * 1e6,
                     "algorithmic_bytes_per_launch": st["algorithmic_bytes_per_sweep"],
                     "layout_bytes_per_launch": st["layout_bytes_per_sweep"],
                     "hbm_stream_gbs_measured": measured_stream_gbs(torch)},
        "sweep_only_msgs_per_s": g.messages_per_sweep() / avg_launch_s,
    }
    if not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(g, ev, a.eps)
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        if ncpu > 1:
            out["cpu_baseline_all_cores"] = cpu_baseline(g, ev, a.eps, budget_s=8.0, threads=min(ncpu, 64))
        ref = cpu_reference_small()
        if ref:
            out["cpu_reference_small"] = ref
    print(json.dumps(out))


if __name__ == "__main__":
    main()

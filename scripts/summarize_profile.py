#!/usr/bin/env python3
"""Condense a scripts/profile_bench.sh output directory into the small files committed under profiles/.

Per workload: the rocprofv3 kernel stats CSV, and -- from the raw kernel trace -- the launches of the
dominant kernel split into STEADY-STATE launches (median and mean reported), the FIRST sweep of each
run (iteration 0 reads no messages) and EARLY EXITS (launches enqueued ahead that found the run already
converged); a plain average over all of them would flatter the kernel.  HBM traffic per launch comes
from the FETCH_SIZE / WRITE_SIZE passes (separate runs) with the gfx950 correction of
MI355X_MICROARCH.md: both counters count KiB, FETCH_SIZE reports half of a wide coalesced read (doubled
here), taken over the same steady-state class.  The summary records the sha256 of the library that was
profiled; bench.py reports `traffic` only while that library is the one running."""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

DOMINANT = {"grid316": ["bp_resident_kernel"], "grid316_launch": ["bp_sweep_kernel"], "dag10k": ["bp_dag_kernel"], "dag10k_launch": ["bp_sweep_kernel"],
            "grid2048": ["bp_sweep_kernel"], "lw": ["lw_sample", "lw_hist"], "lwgen": ["lw_sample", "lw_hist"], "alarm": ["bp_small_kernel"], "mid": ["bp_mid_kernel"],
            "batch_grid316": ["bp_resident_kernel"], "batch_dag10k": ["bp_dag_kernel"]}


def rows_of(path, pattern):
    out = []
    for f in glob.glob(os.path.join(path, "**", pattern), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def classify(trace_rows, kernel):
    """-> list of (class, row) for the launches of `kernel`, in start order."""
    rows = sorted(trace_rows, key=lambda r: int(r["Start_Timestamp"]))
    mine = [r for r in rows if kernel in r["Kernel_Name"]]
    if not mine:
        return []
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mine]
    med = statistics.median(durs)
    out, prev_was_same = [], False
    for r in rows:
        same = kernel in r["Kernel_Name"]
        if same:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if kernel == "bp_sweep_kernel" and d < 0.5 * med:
                cls = "early_exit"
            elif kernel == "bp_sweep_kernel" and not prev_was_same:
                cls = "first_sweep"
            else:
                cls = "steady"
            out.append((cls, r, d))
        prev_was_same = same
    return out


def duration_summary(trace_rows, kernel):
    cl = classify(trace_rows, kernel)
    res = {"launches": len(cl)}
    for cls in ("steady", "first_sweep", "early_exit"):
        d = [x[2] for x in cl if x[0] == cls]
        if d:
            res[cls] = {"count": len(d), "median_ns": statistics.median(d), "mean_ns": sum(d) / len(d)}
    if cl:
        res["mean_all_ns"] = sum(x[2] for x in cl) / len(cl)
    return res


def counter_per_launch(path, kernel, name):
    """median over the steady-state launches of `kernel` of counter `name` (dispatches are matched to the
    trace of the same pass through Dispatch_Id)."""
    cc = [r for r in rows_of(path, "*counter_collection.csv") if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
    if not cc:
        return None
    tr = rows_of(path, "*kernel_trace.csv")
    cls_of = {x[1]["Dispatch_Id"]: x[0] for x in classify(tr, kernel)} if tr else {}
    vals = [float(r["Counter_Value"]) for r in cc if cls_of.get(r["Dispatch_Id"], "steady") == "steady"]
    if not vals:
        vals = [float(r["Counter_Value"]) for r in cc]
    return statistics.median(vals)


def main():
    src, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(dst, exist_ok=True)
    summary = {"tag": tag}
    sha = os.path.join(src, "lib_sha256.txt")
    if os.path.exists(sha):
        summary["lib_sha256"] = open(sha).read().strip()
    failed = os.path.join(src, "failed_passes.txt")
    summary["failed_passes"] = open(failed).read().split("\n")[:-1] if os.path.exists(failed) else []
    for label, kernels in DOMINANT.items():
        tdir = os.path.join(src, "trace_" + label)
        for f in glob.glob(os.path.join(tdir, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(dst, f"{tag}_{label}_kernel_stats.csv"))
            break
        tr = rows_of(tdir, "*kernel_trace.csv")
        traffic = 0.0
        have_traffic = True
        for k in kernels:
            if tr:
                summary[f"{label}_{k}"] = duration_summary(tr, k)
            fetch = counter_per_launch(os.path.join(src, "fetch_" + label), k, "FETCH_SIZE")
            write = counter_per_launch(os.path.join(src, "write_" + label), k, "WRITE_SIZE")
            if fetch is None or write is None:
                have_traffic = False
                continue
            summary[f"{label}_{k}_fetch_size_raw_kib"] = fetch
            summary[f"{label}_{k}_write_size_raw_kib"] = write
            traffic += (2.0 * fetch + write) * 1024.0
        if have_traffic and kernels:
            summary[f"{label}_traffic_bytes_per_launch"] = traffic
        sq = {}
        for sq_dir in (os.path.join(src, "sq_" + label), os.path.join(src, "sq2_" + label)):   # (the SQ counters take two passes)
            if not os.path.isdir(sq_dir):
                continue
            names = sorted({r["Counter_Name"] for r in rows_of(sq_dir, "*counter_collection.csv")})
            for name in names:
                v = counter_per_launch(sq_dir, kernels[0], name)
                if v is not None:
                    sq[name] = v
        if sq:
            summary[f"{label}_sq_counters_per_launch"] = sq
        if label in ("lw", "lwgen"):   # the histogram pass is part of a sample's cost: its counters beside the sampler's
            hq = {}
            for sq_dir in (os.path.join(src, "sq_" + label), os.path.join(src, "sq2_" + label)):
                if os.path.isdir(sq_dir):
                    for name in sorted({r["Counter_Name"] for r in rows_of(sq_dir, "*counter_collection.csv")}):
                        v = counter_per_launch(sq_dir, "lw_hist", name)
                        if v is not None:
                            hq[name] = v
            if hq:
                summary[f"{label}_hist_sq_counters_per_launch"] = hq
        log = os.path.join(src, f"trace_{label}.log")
        if os.path.exists(log):
            lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]
            if lines:
                summary[f"{label}_bench_line"] = json.loads(lines[-1])
                spl = (summary[f"{label}_bench_line"].get("roofline") or {}).get("sweeps_per_launch")
                if spl:
                    summary[f"{label}_sweeps_per_launch"] = spl   # the counters above are per launch: bench.py divides by this
                if label in ("lw", "lwgen"):   # samples one launch of the sample kernel draws (a call is cut into launches of at most 32 GiB of states)
                    import re
                    m = re.search(r"(\d+) samples per call", summary[f"{label}_bench_line"].get("config", {}).get("workload", ""))
                    launches = (summary.get(f"{label}_lw_sample") or {}).get("launches")
                    steps = summary[f"{label}_bench_line"].get("steps")
                    if m and launches and steps:
                        summary[f"{label}_samples_per_launch"] = int(m.group(1)) * (steps + 2) / launches   # (+ 2 warm-up calls)
    vc = os.path.join(src, "valu_clock.txt")
    if os.path.exists(vc):   # scripts/experiments/valu_clock.hip: what the vector ALUs issue on dependent integer code (bench.py: peak_measured_integer_issue)
        rows = [json.loads(ln) for ln in open(vc).read().splitlines() if ln.startswith("{")]
        if rows:
            json.dump({"g_wave_insts_per_s": max(r["G_wave_insts_per_s"] for r in rows), "runs": rows, "tag": tag,
                       "source": "scripts/experiments/valu_clock.hip"}, open(os.path.join(dst, "valu_issue.json"), "w"), indent=1)
    json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if not k.endswith("_bench_line")}, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Condense a scripts/profile_bench.sh output directory into the small files committed under
profiles/: the rocprofv3 kernel stats, and the per-launch HBM traffic of the sweep kernel from the
FETCH_SIZE / WRITE_SIZE passes with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE
reports half of a wide coalesced read: double it; both counters are in KiB... reported as 1e3 B)."""
import csv
import glob
import json
import os
import shutil
import sys


def sweep_counter(path, name):
    vals = []
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "bp_sweep_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    vals.sort()
    # drop first-iteration / early-exit launches: take the median of the upper half
    return vals[len(vals) * 3 // 4] if vals else None


def kernel_stats(path):
    for f in glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True):
        return f
    return None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(dst, exist_ok=True)
    summary = {}
    for suffix, label in (("", "grid316"), ("_2048", "grid2048")):
        ks = kernel_stats(os.path.join(src, "trace" + suffix))
        if ks:
            shutil.copy(ks, os.path.join(dst, f"{tag}_{label}_kernel_stats.csv"))
            for r in csv.DictReader(open(ks)):
                if "bp_sweep_kernel" in r["Name"]:
                    summary[f"{label}_sweep_avg_ns"] = float(r["AverageNs"])
                    summary[f"{label}_sweep_calls"] = int(r["Calls"])
        fetch = sweep_counter(os.path.join(src, "pmc_fetch" + suffix), "FETCH_SIZE")
        write = sweep_counter(os.path.join(src, "pmc_write" + suffix), "WRITE_SIZE")
        if fetch is not None and write is not None:
            # counters are in units of 1 KiB; gfx950: FETCH_SIZE tallies 128-B requests as 64 B
            summary[f"{label}_fetch_size_raw_kib"] = fetch
            summary[f"{label}_write_size_raw_kib"] = write
            summary[f"{label}_traffic_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
    sq = {}
    for f in glob.glob(os.path.join(src, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "bp_sweep_kernel" in r["Kernel_Name"]]
        for name in sorted({r["Counter_Name"] for r in rows}):
            v = sorted(float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name)
            sq[name] = v[len(v) * 3 // 4]
    summary["grid316_sq_counters_per_launch"] = sq
    for log in ("bench_trace.log", "bench_trace_2048.log"):
        p = os.path.join(src, log)
        if os.path.exists(p):
            lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
            if lines:
                summary[log.replace(".log", "_bench_line")] = json.loads(lines[-1])
    json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if not k.endswith("_bench_line")}, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The figures DESIGN.md quotes, regenerated from the committed profiles instead of by hand:
    python scripts/design_figures.py [tag]        (default r06: profiles/<tag>_summary.json + profiles/<tag>_bench_line.json)
Prints one line per figure: kernel time per launch / per sweep (rocprofv3 steady-state mean), PMC traffic per launch and per
second, fractions of the nominal 8 TB/s and of the measured stream rate, VALU share, and the bench extras' headline numbers."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
S = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_summary.json")))
bl_path = os.path.join(ROOT, "profiles", f"{tag}_bench_line.json")
B = json.load(open(bl_path)) if os.path.exists(bl_path) else {}
stream = (B.get("roofline") or {}).get("hbm_stream_gbs_measured")
print(f"profiled library sha256 {S.get('lib_sha256', '?')[:16]}; failed passes: {S.get('failed_passes')}; measured stream {stream} GB/s")


def kernel(label, name):
    d = S.get(f"{label}_{name}") or {}
    return (d.get("steady") or {}).get("mean_ns")


ALG = {"grid316": 89_147_680, "dag10k": 25_354_528, "grid2048": 3_755_999_520}          # SURVEY 8(d) bytes per sweep
MUST = {"grid316": 38_263_808, "dag10k": 3_839_744, "batch_grid316": 38_263_808, "batch_dag10k": 3_839_744}   # messages + node vectors
for label, kern in (("grid316", "bp_resident_kernel"), ("grid316_launch", "bp_sweep_kernel"), ("dag10k", "bp_dag_kernel"),
                    ("dag10k_launch", "bp_sweep_kernel"), ("grid2048", "bp_sweep_kernel"), ("batch_grid316", "bp_resident_kernel"),
                    ("batch_dag10k", "bp_dag_kernel"), ("mid", "bp_mid_kernel"), ("alarm", "bp_small_kernel")):
    ns = kernel(label, kern)
    if not ns:
        continue
    line = f"{label:16s} {kern:20s} {ns / 1e3:9.2f} us per launch"
    spl = S.get(f"{label}_sweeps_per_launch")
    bl = S.get(f"{label}_bench_line") or {}
    if label.startswith("batch_"):
        # a call of 16 sets = four launches of four sets on the resident tiles, one launch of sixteen on the DAG path
        per_call = (bl.get("config") or {}).get("set_sweeps_per_call")
        spl = per_call / (4 if label == "batch_grid316" else 1) if per_call else None
    if spl:
        line += f" = {ns / 1e3 / spl:6.2f} us per {'set-' if label.startswith('batch_') else ''}sweep ({spl:.2f} per launch)"
    t = S.get(f"{label}_traffic_bytes_per_launch")
    if t:
        gbs = t / ns   # bytes per nanosecond = GB/s
        line += f"; traffic {t / 1e6:8.1f} MB per launch = {gbs:6.0f} GB/s = {gbs / 8000:.2f} of 8 TB/s"
        if stream:
            line += f", {gbs / stream:.2f} of the measured stream"
        if spl and label in MUST:
            line += f"; {t / spl / 1e6:.1f} MB per {'set-' if label.startswith('batch_') else ''}sweep = {t / spl / MUST[label]:.2f} x must-move"
    base = label.replace("_launch", "")
    if base in ALG and spl:
        line += f"; SURVEY 8(d) {ALG[base] * spl / ns / 8000:.2f}"
    sq = S.get(f"{label}_sq_counters_per_launch") or {}
    if sq.get("SQ_WAVE_CYCLES"):
        line += f"; VALU active / wave cycle {sq.get('SQ_ACTIVE_INST_VALU', 0) / sq['SQ_WAVE_CYCLES']:.3f}, waiting {sq.get('SQ_WAIT_ANY', 0) / sq['SQ_WAVE_CYCLES']:.2f}"
    if sq.get("SQ_INSTS_VALU") and sq.get("SQ_WAVES") and spl:
        line += f", {sq['SQ_INSTS_VALU'] / sq['SQ_WAVES'] / spl:.0f} VALU instructions per wave and sweep"
    print(line)
for label in ("lw", "lwgen"):
    smp, hst, n = kernel(label, "lw_sample"), kernel(label, "lw_hist"), S.get(f"{label}_samples_per_launch")
    if smp and n:
        sq, hq = S.get(f"{label}_sq_counters_per_launch") or {}, S.get(f"{label}_hist_sq_counters_per_launch") or {}
        line = f"{label:16s} sampler {smp / 1e6:7.2f} ms + histogram {(hst or 0) / 1e6:6.2f} ms per {n:.0f} samples = {n / ((smp + (hst or 0)) * 1e-9):.3e} samples/s in the kernels"
        if sq.get("SQ_INSTS_VALU"):
            line += f"; {sq['SQ_INSTS_VALU'] / n:.0f} + {hq.get('SQ_INSTS_VALU', 0) / n:.0f} VALU wave-instructions per sample"
        t = S.get(f"{label}_traffic_bytes_per_launch")
        if t:
            line += f"; {t / n / 1e3:.1f} KB of HBM traffic per sample"
        print(line)
if B:
    print("bench line:", json.dumps({k: B[k] for k in ("value", "ms_per_step") if k in B}), "roofline", {k: (B.get("roofline") or {}).get(k) for k in ("frac", "frac_survey_8d", "avg_sweep_us", "hbm_stream_gbs_measured", "hbm_stream_gbs_torch_copy")})
    for k in ("batch", "config2_dag", "config5_lw", "grid2048", "config1_alarm", "mid_mixed300"):
        r = B.get(k)
        if not isinstance(r, dict):
            continue
        if k == "batch":
            print("grid batch:", {b: (round(v["value"] / 1e10, 3), round(v["us_per_set_sweep"], 2), round((v.get("cycled") or {}).get("value", 0) / 1e10, 3)) for b, v in r.items() if isinstance(v, dict) and "value" in v}, "(1e10 msgs/s, us per set-sweep, cycled 1e10)")
        elif k == "config2_dag":
            print("configs[1]:", round(r["value"] / 1e9, 3), "e9 cycled;", round(r["same_evidence"]["value"] / 1e9, 3), "e9 same evidence;", "h2h", round(r["value_host_to_host"] / 1e9, 3),
                  "batch", {b: (round(v["value"] / 1e9, 2), round((v.get("cycled") or {}).get("value", 0) / 1e9, 2)) for b, v in r["batch"].items()})
        elif k == "config5_lw":
            g = r.get("generic_mixed10k") or {}
            print("configs[4]:", f"{r['value']:.3e} samples/s, frac", round(r["roofline"].get("frac") or 0, 3), "; generic kernel on mixed10k:", f"{g.get('value', 0):.3e}", "frac", (g.get("roofline") or {}).get("frac"))
        elif k == "grid2048":
            print("grid2048:", f"{r['value']:.3e}", "frac", round(r["roofline"]["frac"], 3))
        elif k == "config1_alarm":
            print("ALARM:", round(r["us_per_query"], 1), "us per query;", {b: round(v["queries_per_s"]) for b, v in r["batch"].items()})
        elif k == "mid_mixed300":
            print("mid300:", round(r["kernel_us_per_sweep"], 2), "us per sweep; mixed10k", round(r["mixed10k"]["kernel_us_per_sweep"], 2))
    d = B.get("dropin_cpp") or {}
    for k in ("config1_alarm", "config2_dag", "config3_grid"):
        if k in d:
            print("dropin", k, {x: d[k][x] for x in ("functor_construct_ms", "functor_construct_again_ms", "flatten_ms", "bn_create_ms", "bn_create_host_ms", "bn_create_device_ms", "c_abi_ms", "run_view_ms", "run_prepared_ms", "operator_ms") if x in d[k]})

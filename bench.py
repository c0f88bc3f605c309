#!/usr/bin/env python3
"""bench.py -- edge-messages/sec to BP convergence on the BASELINE.json grid (config 3).

A "step" is one belief-propagation run to convergence (init, evidence, sweeps until
maximum_difference < eps, beliefs) on the 316x316 k=4 grid BN with 1 % hard evidence, model and
evidence already resident in HBM.  value = 2E * sweeps * steps / wall time, whole job.
One JSON line on stdout (rank 0).  See DESIGN.md "Measurement" for the definitions.

The default single-GPU line also carries, as extra keys measured after the timed region,
  value_host_to_host : SURVEY 8(d)'s definition -- evidence upload to beliefs on the host (PCIe inclusive)
  config1_alarm      : BASELINE configs[0]'s network on the GPU: the ALARM-shaped 37-node net (queries per second)
  mid_mixed300       : a 300-node network of mixed arities (2-5), <= 3 parents: beyond one workgroup's LDS (bn_mid.hip)
                       (+ `mixed10k`: the same path on 10 000 nodes / ~400 k CPT entries, more than 200 workgroups)
  config2_dag        : BASELINE configs[1], the 10 k-node random DAG
  config5_lw         : BASELINE configs[4], likelihood weighting on that DAG
  grid2048           : the HBM-resident point (4.2 M nodes, 3.76 GB per sweep)
  batch              : 2 / 4 / 8 evidence sets per call on the headline grid (bn_bp_run_batch_device)
  dropin_cpp         : the class surface, bn::inference::belief_propagation::operator() / run(), timed in C++ (tests/cpp/bench_dropin.cpp)
each with its own value / roofline / cpu_baseline (--no-extras skips them).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bayesiannetwork_amd import benchline  # noqa: E402  (numpy only: no torch, no library, no GPU call)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def lib_sha256():
    from bayesiannetwork_amd import _lib
    try:
        return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
    except OSError:
        return None


def profiled_traffic(label):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC passes (scripts/profile_bench.sh:
    FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950).  Counters cannot be collected from inside this process, so the number comes from
    the last committed profile of the same command; it is reported only when that profile was taken
    with the library that is running now (sha256 of libbn_mi355x.so), else as stale."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_summary.json"):
            d = json.load(open(os.path.join(pdir, name)))
            if f"{label}_traffic_bytes_per_launch" in d:
                best = (d[f"{label}_traffic_bytes_per_launch"], f"profiles/{name}", d.get("lib_sha256"))
    if not best:
        return {"traffic": None, "traffic_source": None}
    fresh = best[2] is not None and best[2] == lib_sha256()
    return {"traffic": best[0] if fresh else None, "traffic_source": best[1],
            "traffic_stale": None if fresh else best[0]}


def profiled_valu(label, waves_per_simd):
    """VALU-issue fraction of the dominant kernel from the committed SQ counter pass (profiles/*_summary.json,
    `<label>_sq_counters_per_launch`): SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES is the share of a resident wave's
    cycles in which it issues a VALU instruction; times the waves that share a SIMD = how busy the vector ALU
    of an occupied SIMD is.  Like `traffic` it comes from the last committed profile and is marked stale when
    that profile was taken with another build of the library."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_summary.json"):
            d = json.load(open(os.path.join(pdir, name)))
            sq = d.get(f"{label}_sq_counters_per_launch")
            if sq and sq.get("SQ_WAVE_CYCLES"):
                best = (sq, f"profiles/{name}", d.get("lib_sha256"))
    if not best:
        return {"valu_frac": None}
    sq, src, sha = best
    per_wave = sq.get("SQ_ACTIVE_INST_VALU", 0.0) / sq["SQ_WAVE_CYCLES"]
    out = {"valu_frac": min(per_wave * waves_per_simd, 1.0), "valu_active_per_wave_cycle": per_wave,
           "valu_waves_per_simd": waves_per_simd, "valu_source": src,
           "valu_counters": {k: sq[k] for k in ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES") if k in sq}}
    if not (sha is not None and sha == lib_sha256()):
        out["valu_stale"] = True
    return out


def cpu_baseline(model, ev, eps, budget_s=12.0, threads=1, max_sweeps=0, max_runs=8):
    """The oracle (plain-C port of the reference algorithm) timed on this box's host cores on a
    bounded number of runs of the same workload: 1 thread like the reference, or OpenMP over the
    nodes of a sweep (both are reported).  max_sweeps > 0 bounds a run (the 4 M-node
    grid takes ~2 s per sweep on one core)."""
    import oracle
    t0 = time.perf_counter()
    runs, msgs = 0, 0
    while True:
        r = oracle.bp_run(model, ev, eps, threads=threads, max_sweeps=max_sweeps)
        runs += 1
        msgs += model.messages_per_sweep() * r["sweeps"]
        if time.perf_counter() - t0 > budget_s or runs >= max_runs:
            break
    dt = time.perf_counter() - t0
    what = f"{runs} run{'s' if runs > 1 else ''} of the same workload ({r['sweeps']} sweeps each"
    what += ", stopped there by max_sweeps)" if max_sweeps else ")"
    return {"value": msgs / dt, "unit": "edge-messages/s", "cores": threads, "kind": "port",
            "sample": f"{what}, oracle/bp_oracle.c, {threads} thread{'s' if threads > 1 else ''}"}


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


def measured_stream_gbs(torch, device=0):
    """Achievable HBM rate on this box (report against the nominal peak AND a measured stream figure), outside the
    timed region: the library's own streaming kernels (bn_debug_stream: 16 bytes per lane, non-temporal, 1 GiB arrays -- copy and
    triad, best of 5), and torch's device-to-device copy_ of the same size beside them (what rounds 1-5 quoted).
    -> {"copy": GB/s, "triad": GB/s, "torch_copy": GB/s} (a leg that fails is None)."""
    import ctypes
    from bayesiannetwork_amd import _lib
    out = {"copy": None, "triad": None, "torch_copy": None}
    for mode, key in ((0, "copy"), (1, "triad")):
        g = ctypes.c_double(0.0)
        try:
            _lib.check(_lib.lib().bn_debug_stream(device, mode, 1 << 30, 5, ctypes.byref(g)))
            out[key] = g.value
        except Exception:  # noqa: BLE001 - informational only
            pass
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
        out["torch_copy"] = best
    except Exception:  # noqa: BLE001 - informational only
        pass
    return out


def time_bp(eng, g, eps, steps, warmup, torch, event_steps=None):
    """`steps` runs to convergence on the staged evidence, bracketed by synchronize on both sides, with
    nothing but the run itself in the timed region.  Two clocks give the sweep launch duration:
      * the device's own 100 MHz clock, read by the kernels at the first sweep's start and the last
        sweep's end of every timed run (bn_bp_stats.sweep_devclock_ms; free, so it covers the timed region);
      * HIP events the library records on ITS stream around every batch of sweep launches ("timing"
        option; torch.cuda.Event would only see torch's stream).  An event record between two launches
        opens a bubble of several microseconds in the queue, so they are taken on a repeat of the same
        steps right after the timed region instead of inside it."""
    eng.set_option("timing", 0)
    # (untimed, before the W warm-up steps: a quarter of a second of the same run, so that the clocks the timed steps see are the
    # steady ones -- W = 5 steps are 0.7 ms on the headline workload, less than the power state takes to settle after an idle spell)
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.25 and os.environ.get("BN_BENCH_NO_SETTLE") != "1":   # (scripts/profile_bench.sh: every dispatch is a row of its traces)
        eng.bp_run_device(eps)
    for _ in range(max(warmup, 1)):
        r = eng.bp_run_device(eps)
    torch.cuda.synchronize()
    sweeps_total, dev_ms = 0, 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        r = eng.bp_run_device(eps)
        st = eng.bp_stats()
        sweeps_total += r["sweeps"]
        dev_ms += st["sweep_devclock_ms"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    path = eng.last_path()
    eng.set_option("timing", 1)
    ev_steps = event_steps if event_steps is not None else min(steps, 20)
    kern_ms, launches, ev_sweeps = 0.0, 0, 0
    eng.bp_run_device(eps)
    t1 = time.perf_counter()
    for _ in range(ev_steps):
        r = eng.bp_run_device(eps)
        st = eng.bp_stats()
        kern_ms += st["sweep_kernel_ms"]
        launches += st["sweep_launches"]
        ev_sweeps += r["sweeps"]
    torch.cuda.synchronize()
    dt_ev = time.perf_counter() - t1
    eng.set_option("timing", 0)
    st = eng.bp_stats()
    # per-sweep launches: one launch = one sweep.  One-launch paths: a launch is the whole run; the
    # per-sweep figure divides its duration by the sweeps it executed (same bytes-per-second either way).
    avg_sweep_s = kern_ms * 1e-3 / max(ev_sweeps, 1)
    avg_launch_s = kern_ms * 1e-3 / max(launches, 1)
    avg_dev_s = dev_ms * 1e-3 / max(sweeps_total, 1)
    if avg_sweep_s <= 0:
        avg_sweep_s = avg_dev_s
    achieved = st["algorithmic_bytes_per_sweep"] / avg_sweep_s / 1e9
    return {"dt": dt, "sweeps_total": sweeps_total, "msgs": g.messages_per_sweep() * sweeps_total,
            "avg_sweep_s": avg_sweep_s, "avg_launch_s": avg_launch_s, "sweeps_per_launch": ev_sweeps / max(launches, 1),
            "avg_sweep_devclock_s": avg_dev_s, "achieved": achieved, "stats": st,
            "path": path, "ms_per_step_with_events": dt_ev / max(ev_steps, 1) * 1e3, "event_steps": ev_steps,
            "n_tiles": eng.layout()["n_tiles"], "cpt_bytes": 8 * len(g.cpt),
            "waves_per_block": {2: eng.info("resident_waves"), 5: 8}.get(path)}


PATH_KERNEL = {0: "bp_sweep_kernel", 2: "bp_resident_kernel", 3: "bp_small_kernel", 4: "bp_mid_kernel", 5: "bp_dag_kernel"}
PATH_NAME = {0: "one launch per sweep", 2: "resident tiles, one launch for the whole run (grid barrier per sweep)",
             3: "one workgroup, state in LDS, one launch for the whole run (small networks)",
             4: "the same items over several workgroups, state in memory, grid barrier per iteration, one launch for the whole run (mid-size networks)",
             5: "child tiles with the CPT in registers + parent items on waves of their own, state in memory, grid barrier per iteration, "
                "one launch for the whole run (networks of arity <= 4 with <= 5 parents)"}


ENGINE_CLOCK_GHZ = 2.4  # MI355X peak engine clock, /opt/skills/guides/MI355X_MICROARCH.md
VALU_CYCLES_PER_INST = 4  # a 64-lane wavefront issues one vector instruction over 4 cycles of its 16-lane SIMD


def resident_record(t, label, waves_per_simd):
    """One-launch paths that keep the CPTs on chip (resident tiles, register-resident DAG path): the bound SURVEY 8(d)'s figure
    cannot give them.  Per sweep such a kernel MUST still move the messages and node vectors (read once, written once:
    algorithmic bytes minus the CPT term) -> floor at the HBM peak; and it MUST issue its vector instructions (SQ_INSTS_VALU of
    the committed counter pass, per wave and sweep, x 4 cycles x the waves that share a SIMD) -> VALU-issue floor.
    frac_resident = max(floors) / measured sweep time <= 1; `bound` names the floor that binds."""
    st = t["stats"]
    must_move = st["algorithmic_bytes_per_sweep"] - t["cpt_bytes"]
    sweep_us = t["avg_sweep_s"] * 1e6
    floor_hbm_us = must_move / (HBM_PEAK_GBS * 1e9) * 1e6
    out = {"bytes_per_sweep": must_move, "what": "pi-/lambda-messages and node vectors, read once and written once per sweep (the CPTs stay on chip)",
           "floor_hbm_us": floor_hbm_us, "measured_sweep_us": sweep_us, "achieved_gbs": must_move / max(t["avg_sweep_s"], 1e-12) / 1e9}
    floor_valu_us = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_summary.json"):
            d = json.load(open(os.path.join(pdir, name)))
            sq = d.get(f"{label}_sq_counters_per_launch") or {}
            kern = next((v for k, v in d.items() if k.startswith(label + "_bp_") and isinstance(v, dict) and "steady" in v), None)
            if sq.get("SQ_INSTS_VALU") and sq.get("SQ_WAVES") and d.get(f"{label}_sweeps_per_launch"):
                per_wave_sweep = sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"] / d[f"{label}_sweeps_per_launch"]
                floor_valu_us = per_wave_sweep * VALU_CYCLES_PER_INST * waves_per_simd / (ENGINE_CLOCK_GHZ * 1e3)
                out.update({"valu_insts_per_wave_sweep": per_wave_sweep, "valu_source": f"profiles/{name}",
                            "valu_stale": not (d.get("lib_sha256") is not None and d.get("lib_sha256") == lib_sha256()),
                            "waves_per_simd": waves_per_simd, "engine_clock_ghz": ENGINE_CLOCK_GHZ})
            del kern
    out["floor_valu_us"] = floor_valu_us
    floor = max(floor_hbm_us, floor_valu_us or 0.0)
    out["bound"] = "valu" if (floor_valu_us or 0.0) > floor_hbm_us else "hbm"
    out["frac_resident"] = floor / max(sweep_us, 1e-9)
    return out


def roofline_of(t, label):
    """The dominant kernel against the bound that applies to it.  Scalars first, nested records last (a reader that keeps
    only the leading scalars of this object must still see the figure that means something).

    One launch per sweep (path 0): SURVEY 8(d) -- achieved = algorithmic bytes of a sweep / the launch's duration, against the
    HBM peak.  One-launch paths that keep the CPTs on chip (2: resident tiles, 5: register-resident DAG): SURVEY 8(d)'s figure
    prices a CPT read per sweep these kernels do not perform (it reaches 1.0 while the memory system moves a third of those
    bytes), so `frac` is frac_resident -- the larger of (message + node-vector bytes that must still move, at the HBM peak) and
    (the vector instructions that must issue) over the measured sweep time -- and the 8(d) figure stays as frac_survey_8d."""
    st = t["stats"]
    spl = t["sweeps_per_launch"]
    achieved_8d = t["achieved"]
    out = {"bound": "hbm", "achieved": achieved_8d, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_8d / HBM_PEAK_GBS}
    traffic = profiled_traffic(label)
    out["traffic"] = traffic["traffic"]
    out["frac_survey_8d"] = achieved_8d / HBM_PEAK_GBS
    out["achieved_survey_8d"] = achieved_8d
    nested = {}
    if t["path"] in (2, 5):
        # resident tiles: 8 waves per 512-thread block on 4 SIMDs
        wps = (t["waves_per_block"] or 8) / 4.0
        res = resident_record(t, label, wps)
        out["bound"] = res["bound"]
        out["frac"] = out["frac_resident"] = res["frac_resident"]
        if res["bound"] == "hbm":
            out["achieved"] = res["achieved_gbs"]
        else:   # vector issue: wave-instructions per second over what the occupied SIMDs can issue, scaled to the chip
            out["unit"] = "G wave-instructions/s"
            out["peak"] = 256 * 4 * ENGINE_CLOCK_GHZ / VALU_CYCLES_PER_INST
            out["achieved"] = out["peak"] * res["frac_resident"]
        out["floor_hbm_us"] = res["floor_hbm_us"]
        out["floor_valu_us"] = res["floor_valu_us"]
        out["must_move_bytes_per_sweep"] = res["bytes_per_sweep"]
        out["limiter"] = "latency: barrier hand-off + dependent message round trips above the " + res["bound"] + " floor"
        out["note"] = ("frac = frac_resident = max(message + node-vector bytes at the HBM peak, VALU issue time) / measured sweep time; "
                       "frac_survey_8d = SURVEY 8(d) algorithmic bytes (a CPT read per node and sweep) / time, kept for comparison with "
                       "the per-sweep formulation: this kernel keeps the CPTs in registers, so that figure is not a roofline for it")
        nested["resident"] = res
    out.update({"kernel": PATH_KERNEL.get(t["path"], "?"),
                "avg_launch_us": t["avg_launch_s"] * 1e6, "sweeps_per_launch": spl,
                "avg_sweep_us": t["avg_sweep_s"] * 1e6,
                "avg_launch_us_source": f"HIP events on the engine's stream, {t['event_steps']} repeated steps after the timed region "
                                        f"({t['ms_per_step_with_events']:.4f} ms per step with the events in the queue)",
                "avg_sweep_us_devclock": t["avg_sweep_devclock_s"] * 1e6,
                "achieved_devclock": st["algorithmic_bytes_per_sweep"] / max(t["avg_sweep_devclock_s"], 1e-12) / 1e9,
                "algorithmic_bytes_per_launch": st["algorithmic_bytes_per_sweep"] * spl,
                "algorithmic_bytes_per_sweep": st["algorithmic_bytes_per_sweep"],
                "layout_bytes_per_sweep": st["layout_bytes_per_sweep"],
                "traffic_source": traffic["traffic_source"], "traffic_stale": traffic.get("traffic_stale")})
    if out.get("traffic"):
        out["traffic_gbs"] = out["traffic"] / max(t["avg_launch_s"], 1e-12) / 1e9  # what the memory system actually moved
    # resident tiles: 8 waves per 512-thread block on 4 SIMDs; per-sweep launches: 256-thread blocks, 2 blocks per CU
    valu = profiled_valu(label, 2 if t["path"] == 2 else min(2.0, max(1.0, t["n_tiles"] / 1024.0)))
    if "valu_counters" in valu:
        nested["valu_counters"] = valu.pop("valu_counters")
    out.update(valu)
    out.update(nested)
    return out


def evidence_cycle(g, frac, n=8):
    """n evidence sets of the same size on different nodes: a stream of different queries, so that the sweep
    count of the next run is not the one the engine has just seen (its launch path enqueues the predicted
    number of sweeps ahead)."""
    from bayesiannetwork_amd import synth
    return [synth.random_evidence(g, frac, seed=7 + q) for q in range(n)]


def time_host_to_host(eng, g, evs, eps, steps):
    """wall time from evidence upload to beliefs on the host, per query, over a cycle of different
    evidence sets (bn_bp_run_view = evidence H2D, evidence kernel, run, 8 * sum(k) bytes of beliefs D2H into the
    engine's page-locked buffer, ONE synchronisation); PCIe inclusive.  SURVEY 8(d) defines the metric's `t` this way:
    reported at the top level of the line as value_host_to_host / frac_host_to_host beside the device-resident `value`."""
    # warm-up by TIME: for some tens of milliseconds after `import torch` a process answers short queries 2-3x slower (measured:
    # ALARM-sized network 157 vs 58 us per query in the first 400 queries, scripts/experiments/alarm_torch.py)
    t0 = time.perf_counter()
    i = 0
    while i < 2 or (time.perf_counter() - t0 < 0.15 and i < 4096):
        eng.bp_run_view(evs[i % len(evs)], eps)
        i += 1
    # per-call clock, MEDIAN over the calls (this process also hosts torch's and the profiler's threads: the mean of 20 calls moved by
    # +-10 % from run to run, 0.227-0.242 ms, where the same loop alone in a process gives 0.202 +- 0.002 --
    # scripts/experiments/h2h_probe.py); the mean is reported beside it
    # ONE statistic: the median over the calls of (messages of call i) / (time of call i) -- the sets of the cycle need different sweep counts
    steps = max(steps, 96)
    sweeps, per_call, rates = 0, [], []
    for i in range(steps):
        t0 = time.perf_counter()
        sw = eng.bp_run_view(evs[i % len(evs)], eps)["sweeps"]
        dt = time.perf_counter() - t0
        sweeps += sw
        per_call.append(dt)
        rates.append(g.messages_per_sweep() * sw / dt)
    per_call.sort()
    rates.sort()
    med = per_call[len(per_call) // 2]
    out = {"value": rates[len(rates) // 2], "unit": "edge-messages/s", "ms_per_step": med * 1e3,
           "ms_per_step_mean": sum(per_call) / steps * 1e3, "steps": steps, "evidence_sets_cycled": len(evs), "sweeps_per_step": sweeps / steps,
           "what": "bn_bp_run_view: evidence H2D + run to convergence + beliefs D2H (pinned), one sync, host wall clock; value = median over the "
                   "calls of messages_i / t_i, ms_per_step = median t_i"}
    # the plain entry point with a caller-owned (pageable) array, for comparison
    eng.bp_run(evs[0], eps)
    t0 = time.perf_counter()
    sw2 = 0
    for i in range(steps):
        sw2 += eng.bp_run(evs[i % len(evs)], eps)["sweeps"]
    dt2 = time.perf_counter() - t0
    out["pageable_out"] = {"value": g.messages_per_sweep() * sw2 / dt2, "ms_per_step": dt2 / steps * 1e3,
                           "what": "bn_bp_run into a caller-owned pageable array"}
    return out


def time_cycled(eng, g, evs, eps, steps):
    """Device-resident runs over a cycle of DIFFERENT evidence sets: each set is staged by bn_bp_set_evidence outside
    the clock, the clock runs around bn_bp_run_device only (the call returns after its own synchronisation).  On
    the launch path the engine enqueues the sweep count of the PREVIOUS run ahead, so a stream of different queries
    pays for over- and under-shoots that a repeat of the same query never sees."""
    t0 = time.perf_counter()
    i = 0
    while i < 2 or (time.perf_counter() - t0 < 0.1 and i < 4096):   # warm-up by time (see time_host_to_host)
        eng.bp_set_evidence(evs[i % len(evs)])
        eng.bp_run_device(eps)
        i += 1
    blocks = []
    for _ in range(3):   # three blocks of `steps` runs, the median block is reported (a leg that follows seconds of CPU-only work
        dt, sweeps, launches = 0.0, 0, 0   # -- another leg's CPU baseline -- has been seen 70 % slow for a whole block)
        for i in range(steps):
            eng.bp_set_evidence(evs[i % len(evs)])
            t0 = time.perf_counter()
            r = eng.bp_run_device(eps)
            dt += time.perf_counter() - t0
            sweeps += r["sweeps"]
            launches += eng.bp_stats()["sweep_launches"]
        blocks.append((dt, sweeps, launches))
    blocks.sort()
    dt, sweeps, launches = blocks[1]
    return {"value": g.messages_per_sweep() * sweeps / dt, "unit": "edge-messages/s", "ms_per_step": dt / steps * 1e3,
            "steps": steps, "evidence_sets_cycled": len(evs), "sweeps_per_step": sweeps / steps,
            "sweep_launches_per_step": launches / steps, "ms_per_step_blocks": [round(b[0] / steps * 1e3, 4) for b in blocks],
            "what": "bn_bp_run_device on a cycle of different staged evidence sets; clock around the run only; median of three blocks"}


def wsum64(arr):
    """Order-sensitive checksum tests/cpp/bench_dropin.cpp states in C++: sum_i word_i * (2 i + 1) mod 2^64."""
    import numpy as np
    w = np.ascontiguousarray(arr, dtype=np.float64).view(np.uint64)
    with np.errstate(over="ignore"):
        return int((w * (np.arange(w.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64))


def leg_dropin_cpp(a, local_rank, torch):
    """The drop-in where the reference's user calls it: bn::inference::belief_propagation::operator()(precondition, epsilon)
    (reference belief_propagation.hpp:31-159) timed by a C++ program over include/compat's graph_t (tests/cpp/bench_dropin.cpp,
    plain g++), on BASELINE configs[0], [1], [2], with its split -- evidence marshal / bn_bp_run_view / building the reference's
    unordered_map<vertex_type, matrix_type> / the caller destroying it -- beside run(), the same query through the non-owning
    marginals_view.  A child process, outside every timed region of this script; its sweeps and a checksum of query 0's
    marginals are compared with this process' own run of the same network and evidence through ctypes."""
    import subprocess

    import __graft_entry__
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.dsc import load_dsc
    from bayesiannetwork_amd.engine import Engine
    exe = __graft_entry__.build_bench_dropin()
    env = dict(os.environ)
    env.setdefault("HIP_VISIBLE_DEVICES", str(local_rank))
    p = subprocess.run([exe, "--dsc", os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc")], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    if p.returncode != 0:
        return {"error": f"bench_dropin rc {p.returncode}: {p.stderr[-300:]}"}
    out = json.loads(p.stdout.strip().splitlines()[-1])
    nets = {"config1_alarm": (lambda: load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))[0], 0.1, 1e-6),
            "config2_dag": (lambda: synth.random_dag(10000, 4, 64, 4, seed=1), 0.01, 1e-3),
            "config3_grid": (lambda: synth.grid(316, 316, 4, seed=2), 0.01, 1e-3)}
    for key, (make, frac, eps) in nets.items():
        if key not in out:
            continue
        g = make()
        with Engine(g, device=local_rank) as eng:
            r = eng.bp_run(synth.random_evidence(g, frac, seed=7), eps)   # (a copy: bp_run_view's array dies with the engine)
        out[key]["matches_c_abi"] = bool(r["sweeps"] == out[key]["sweeps_query0"]
                                         and f"{wsum64(r['beliefs']):016x}" == out[key]["wsum64_query0"])
    out["what"] = ("tests/cpp/bench_dropin.cpp: medians per query, ms; operator_ms = bn::inference::belief_propagation::operator() called, "
                   "its map used and dropped; run_view_ms = run() (marshal + bn_bp_run_view, marginals read in place); run_prepared_ms = run() on evidence prepared "
                   "once (prepare(): no walk over the caller's map); map_build_copy_assign_ms = "
                   "how the map was built before this round (default-constructed entry + copy assignment per node)")
    return out


def leg_dag(a, local_rank, torch):
    """BASELINE configs[1]: Loopy BP on the 10 k-node random DAG (<= 4 parents, k = 4)."""
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import Engine
    g = synth.random_dag(10000, 4, 64, 4, seed=1)
    ev = synth.random_evidence(g, a.evidence, seed=7)
    with Engine(g, device=local_rank) as eng:
        eng.bp_set_evidence(ev)
        t = time_bp(eng, g, a.eps, max(a.steps, 20), a.warmup, torch)
        evs = evidence_cycle(g, a.evidence)
        cyc = time_cycled(eng, g, evs, a.eps, max(a.steps, 24))
        h2h = time_host_to_host(eng, g, evs, a.eps, 16)
        run_path = eng.last_path()
        # B queries per call: every per-sweep launch carries all sets (blockIdx.y), so they share its latency
        batch = time_batches(eng, g, a, torch, (4, 16))
        eng.set_option("dag", 0)   # the tile kernels on the same network: one launch per sweep
        eng.bp_set_evidence(ev)
        tl = time_bp(eng, g, a.eps, 20, 3, torch, event_steps=10)
        eng.set_option("dag", 1)
    steps = max(a.steps, 20)
    out = {"workload": f"10 k-node random DAG, <=4 parents, k=4, {g.n_edges} edges (BASELINE.json configs[1]), "
                       f"{ev.ne} evidence nodes, eps={a.eps:g}",
           "value": cyc["value"], "unit": "edge-messages/s", "ms_per_step": cyc["ms_per_step"], "steps": cyc["steps"],
           "sweeps_per_step": cyc["sweeps_per_step"], "messages_per_sweep": g.messages_per_sweep(),
           "cycled_evidence": cyc,
           "same_evidence": {"value": t["msgs"] / t["dt"], "ms_per_step": t["dt"] / steps * 1e3, "steps": steps,
                             "sweeps_per_step": t["sweeps_total"] / steps,
                             "what": "the same staged evidence set run again and again (the sweep prediction is then exact)"},
           "value_host_to_host": h2h["value"], "host_to_host": h2h, "run_path": PATH_NAME.get(run_path),
           "roofline": roofline_of(t, "dag10k" if t["path"] == 5 else "dag10k_launch"),
           "tile_kernels": {"path": PATH_NAME.get(tl["path"]), "ms_per_step": tl["dt"] / 20 * 1e3, "avg_sweep_us": tl["avg_sweep_s"] * 1e6,
                            "frac": tl["achieved"] / HBM_PEAK_GBS}, "batch": batch}
    if not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(g, ev, a.eps, budget_s=6.0)
    # the same path on a network of MIXED arities (2..4, padded to 4 in registers; <= 4 parents, 10 000 nodes, 723 k CPT entries --
    # beyond the item kernels): device-resident runs on a cycle of evidence sets, the tile kernels beside it
    gm = synth.random_dag(10000, 4, 64, [2, 3, 4], seed=8)
    evm = evidence_cycle(gm, a.evidence)
    with Engine(gm, device=local_rank) as eng:
        cm = time_cycled(eng, gm, evm, a.eps, 16)
        pm = eng.last_path()
        eng.set_option("dag", 0)
        c0 = time_cycled(eng, gm, evm, a.eps, 8)
        p0 = eng.last_path()
    out["mixed_arity_10k"] = {"workload": f"10 k-node random DAG, arities 2-4, <=4 parents, {gm.n_edges} edges, {int(gm.cpt_off[-1])} CPT entries",
                              "value": cm["value"], "unit": "edge-messages/s", "ms_per_step": cm["ms_per_step"], "sweeps_per_step": cm["sweeps_per_step"],
                              "run_path": PATH_NAME.get(pm),
                              "without_this_path": {"value": c0["value"], "ms_per_step": c0["ms_per_step"], "run_path": PATH_NAME.get(p0)}}
    return out


def leg_alarm(a, local_rank, torch):
    """BASELINE configs[0]'s network (ALARM-shaped: 37 nodes, mixed arities, up to 4 parents -- tests/golden/alarm_shaped.dsc)
    through the drop-in's host path: what a user of the reference pays per query.  The whole run is one workgroup with
    the state in LDS (csrc/bn_small.hip); a batch runs one workgroup per evidence set."""
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.dsc import load_dsc
    from bayesiannetwork_amd.engine import Engine
    g, _ = load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))
    eps = 1e-6
    evs = [synth.random_evidence(g, 0.1, seed=7 + q) for q in range(8)]
    out = {"workload": f"ALARM-shaped network: {g.n} nodes, {g.n_edges} edges, {len(g.cpt)} CPT entries, 10 % evidence on different nodes "
                       f"per query, eps={eps:g}"}
    with Engine(g, device=local_rank) as eng:
        steps = 400
        h2h = time_host_to_host(eng, g, evs, eps, steps)
        out["path"] = PATH_NAME.get(eng.last_path())
        dev = 0.0
        sweeps = 0
        for i in range(64):
            r = eng.bp_run_view(evs[i % len(evs)], eps)
            dev += eng.bp_stats()["sweep_devclock_ms"]
            sweeps += r["sweeps"]
        out.update({"value": 1e3 / h2h["ms_per_step"], "unit": "queries/s (evidence in, marginals on the host, one query per call)",
                    "us_per_query": h2h["ms_per_step"] * 1e3, "sweeps_per_query": h2h["sweeps_per_step"],
                    "edge_messages_per_s": h2h["value"], "kernel_us_per_sweep": dev / sweeps * 1e3, "host_to_host": h2h})
        eng.set_option("small", 0)   # the tile kernels on the same network: one launch per sweep
        eng.set_option("multisweep", 0)
        tiles = time_host_to_host(eng, g, evs, eps, 100)
        out["tile_kernels"] = {"us_per_query": tiles["ms_per_step"] * 1e3, "path": PATH_NAME.get(eng.last_path())}
        eng.set_option("small", 1)
        eng.set_option("multisweep", 1)
        batch = {}
        for B in (16, 64, 256):
            sets = [synth.random_evidence(g, 0.1, seed=100 + q) for q in range(B)]
            eng.bp_set_evidence_batch(sets)
            for _ in range(3):
                eng.bp_run_batch_device(eps)
            reps = 50
            t0 = time.perf_counter()
            sw = 0
            for _ in range(reps):
                sw += int(eng.bp_run_batch_device(eps)["sweeps"].sum())
            dt = time.perf_counter() - t0
            batch[f"B{B}"] = {"queries_per_s": B * reps / dt, "us_per_call": dt / reps * 1e6, "us_per_set_sweep": dt / sw * 1e6,
                              "path": PATH_NAME.get(eng.last_path()), "what": "bn_bp_run_batch_device: evidence staged, marginals left in HBM"}
            # the whole call: the caller's evidence arrays in, every set's marginals in the caller's array out
            import ctypes
            import numpy as np
            from bayesiannetwork_amd import _lib
            from bayesiannetwork_amd.engine import _p
            ne, node, off, val = eng._pack_sets(sets)
            bel = np.empty((B, int(g.k.sum())))
            sweeps_out, res_out = np.zeros(B, dtype=np.int32), np.zeros(B)
            L = _lib.lib()

            def call():
                _lib.check(L.bn_bp_run_batch(eng._h, B, _p(ne, ctypes.c_int32), _p(node, ctypes.c_int32), _p(off, ctypes.c_int32),
                                             _p(val, ctypes.c_double), eps, 0, _p(bel, ctypes.c_double), _p(sweeps_out, ctypes.c_int32),
                                             _p(res_out, ctypes.c_double)))
            for _ in range(5):
                call()
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            dt = time.perf_counter() - t0
            batch[f"B{B}"]["end_to_end"] = {"queries_per_s": B * reps / dt, "us_per_call": dt / reps * 1e6,
                                            "what": "bn_bp_run_batch: evidence arrays in, marginals in the caller's array out"}
        out["batch"] = batch
    if not a.no_cpu:
        import oracle
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.0:
            oracle.bp_run(g, evs[n % len(evs)], eps)
            n += 1
        out["cpu_baseline"] = {"value": n / (time.perf_counter() - t0), "unit": "queries/s", "cores": 1, "kind": "port",
                               "sample": f"{n} queries of the same cycle through oracle/bp_oracle.c, 1 thread"}
        ref = None
        try:
            if oracle.ref_available():
                r = oracle.ref_bp(g, evs[0], eps, timeout=60)
                ref = {"value": 1.0 / r["sweep_s"] if r["sweep_s"] > 0 else None, "unit": "queries/s", "cores": 1, "kind": "reference",
                       "sample": f"one query, {r['sweeps']} sweeps, bn::inference::belief_propagation unmodified (time of the while(true) loop only)"}
        except Exception as ex:  # noqa: BLE001 - informational
            ref = {"error": str(ex)[:200]}
        if ref:
            out["cpu_reference"] = ref
    return out


def leg_mid(a, local_rank, torch):
    """A mid-size network of mixed arities (not a BASELINE config; the kind between configs[0] and configs[1]): 300 nodes, arities
    2-5, up to 3 parents, 10.6 k CPT entries -- the item kernel over 37 workgroups (csrc/bn_mid.hip) beside the tile kernels."""
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import Engine
    g = synth.random_dag(300, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=12)
    eps = 1e-6
    evs = [synth.random_evidence(g, 0.05, seed=7 + q) for q in range(8)]
    out = {"workload": f"{g.n}-node random DAG, arities 2-5, <= 3 parents, {g.n_edges} edges, {len(g.cpt)} CPT entries, 5 % evidence on different "
                       f"nodes per query, eps={eps:g}"}
    with Engine(g, device=local_rank) as eng:
        h2h = time_host_to_host(eng, g, evs, eps, 200)
        out["path"] = PATH_NAME.get(eng.last_path())
        out["workgroups"] = eng.info("mid_parts")
        dev, sweeps = 0.0, 0
        for i in range(32):
            r = eng.bp_run_view(evs[i % len(evs)], eps)
            dev += eng.bp_stats()["sweep_devclock_ms"]
            sweeps += r["sweeps"]
        out.update({"value": h2h["value"], "unit": "edge-messages/s (evidence in, marginals on the host, one query per call)",
                    "us_per_query": h2h["ms_per_step"] * 1e3, "sweeps_per_query": h2h["sweeps_per_step"], "kernel_us_per_sweep": dev / sweeps * 1e3})
        eng.set_option("mid", 0)
        tiles = time_host_to_host(eng, g, evs, eps, 100)
        out["tile_kernels"] = {"us_per_query": tiles["ms_per_step"] * 1e3, "path": PATH_NAME.get(eng.last_path())}
        eng.set_option("mid", 1)
        sets = [synth.random_evidence(g, 0.05, seed=100 + q) for q in range(64)]
        eng.bp_set_evidence_batch(sets)
        for _ in range(3):
            eng.bp_run_batch_device(eps)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.bp_run_batch_device(eps)
        dt = time.perf_counter() - t0
        out["batch_B64"] = {"queries_per_s": 64 * reps / dt, "us_per_call": dt / reps * 1e6, "path": PATH_NAME.get(eng.last_path())}
    if not a.no_cpu:
        import oracle
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.0:
            oracle.bp_run(g, evs[n % len(evs)], eps)
            n += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / dt, "unit": "queries/s", "cores": 1, "kind": "port",
                               "sample": f"{n} queries of the same cycle through oracle/bp_oracle.c, 1 thread"}
    # the same path at the scale of configs[1]: 10 000 nodes of mixed arities, ~400 k CPT entries, more than 200 workgroups
    big = synth.random_dag(10000, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=20)
    evs = [synth.random_evidence(big, 0.01, seed=7 + q) for q in range(8)]
    with Engine(big, device=local_rank) as eng:
        h2h = time_host_to_host(eng, big, evs, eps, 50)
        o10 = {"workload": f"{big.n}-node random DAG, arities 2-5, <= 3 parents, {big.n_edges} edges, {len(big.cpt)} CPT entries, 1 % evidence, eps={eps:g}",
               "path": PATH_NAME.get(eng.last_path()), "workgroups": eng.info("mid_parts"), "value": h2h["value"],
               "unit": "edge-messages/s (evidence in, marginals on the host, one query per call)", "us_per_query": h2h["ms_per_step"] * 1e3,
               "sweeps_per_query": h2h["sweeps_per_step"]}
        dev, sweeps = 0.0, 0
        for i in range(16):
            r = eng.bp_run_view(evs[i % len(evs)], eps)
            dev += eng.bp_stats()["sweep_devclock_ms"]
            sweeps += r["sweeps"]
        o10["kernel_us_per_sweep"] = dev / sweeps * 1e3
        eng.set_option("mid", 0)
        tiles = time_host_to_host(eng, big, evs, eps, 30)
        o10["tile_kernels"] = {"us_per_query": tiles["ms_per_step"] * 1e3, "path": PATH_NAME.get(eng.last_path())}
    out["mixed10k"] = o10
    return out


def measured_valu_issue_peak():
    """G wave-instructions/s this chip sustains on dependent integer code under load (scripts/experiments/valu_clock.hip), read from the
    committed experiment output (profiles/valu_issue.json) rather than quoted as a literal; None when absent."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "valu_issue.json")))["g_wave_insts_per_s"]
    except (OSError, KeyError, ValueError):
        return None


def leg_lw(a, local_rank, torch, generic=False):
    """BASELINE configs[4] on one GPU: weighted samples/s on the 10 k-node DAG with 1 % evidence (the straight-line sampling kernel:
    every node <= 4 states, <= 4 parents, <= 256 rows).  generic=True: the GENERIC sampling kernel (lw_sample_kernel + lw_hist_kernel)
    on `mixed10k` -- 10 000 nodes of arities 2-5, outside the straight-line kernel's domain -- 10^6 samples per call
    (reference likelihood_weighting.hpp:122-193)."""
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import Engine
    if generic:
        d = synth.random_dag(10000, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=20)
        samples, label = min(a.samples, 1000000), "lwgen"
    else:
        d = synth.random_dag(10000, 4, 64, 4, seed=1)
        samples, label = a.samples, "lw"
    ev = synth.random_evidence(d, a.evidence, seed=7).hard_states(d)
    steps = max(1, min(a.steps, 5))
    with Engine(d, device=local_rank) as eng:
        for w in range(2):
            eng.lw_run(ev, samples, seed=1, sample_begin=w * samples)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.lw_run(ev, samples, seed=1, sample_begin=(i + 2) * samples)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        small_kernel = bool(eng.info("lw_small"))
    assert small_kernel != generic, "the sampling kernel that ran is not the one this leg is about"
    rate = samples * steps / dt
    # What bounds the sampler: not HBM (the CPTs are cache-resident) but the vector ALU and the row gathers.  The
    # bound reported is VALU ISSUE: vector instructions per second (SQ_INSTS_VALU of the committed counter pass, per sample, x the
    # measured sample rate) against what the chip can issue (CUs x 4 SIMDs x clock / 4 cycles per 64-lane instruction).
    bytes_per_sample = (d.n * 2 + d.n_edges) // (4 if small_kernel else 1)   # state written, parents' states read, re-read by the histogram pass (informational); two bits per state on the straight-line kernel
    peak_ginst = 256 * 4 * ENGINE_CLOCK_GHZ / VALU_CYCLES_PER_INST    # G wave-instructions / s
    roof = {"bound": "valu", "peak": peak_ginst, "unit": "G wave-instructions/s", "achieved": None, "frac": None,
            "kernel": ("lw_sample_small_kernel + lw_hist2_kernel" if small_kernel else "lw_sample_kernel + lw_hist_kernel"),
            "hbm_algorithmic_bytes_per_sample": bytes_per_sample, "hbm_algorithmic_gbs": rate * bytes_per_sample / 1e9,
            "peak_measured_integer_issue": measured_valu_issue_peak(),
            "note": "frac = vector instructions issued per second (SQ_INSTS_VALU per sample of the committed SQ pass, sampling + histogram kernels, x measured "
                    "samples/s) over one instruction per SIMD and four cycles at 2.4 GHz (DESIGN.md section 4.7)"}
    roof.update(profiled_traffic(label))
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_summary.json"):
            dd = json.load(open(os.path.join(pdir, name)))
            sq = dd.get(f"{label}_sq_counters_per_launch") or {}
            per_launch = dd.get(f"{label}_samples_per_launch")
            hq = dd.get(f"{label}_hist_sq_counters_per_launch") or {}
            if sq.get("SQ_INSTS_VALU") and per_launch:
                inst_per_sample = (sq["SQ_INSTS_VALU"] + hq.get("SQ_INSTS_VALU", 0.0)) / per_launch   # (both kernels run once per launch of samples)
                roof["valu_insts_per_sample_sampler"] = sq["SQ_INSTS_VALU"] / per_launch
                roof["valu_insts_per_sample_histogram"] = hq.get("SQ_INSTS_VALU", 0.0) / per_launch
                roof.update({"valu_insts_per_sample": inst_per_sample, "valu_insts_per_node_sample_lane": inst_per_sample * 64 / d.n,
                             "achieved": rate * inst_per_sample / 1e9, "frac": rate * inst_per_sample / 1e9 / peak_ginst,
                             "valu_source": f"profiles/{name}",
                             "valu_stale": not (dd.get("lib_sha256") is not None and dd.get("lib_sha256") == lib_sha256())})
    what = (f"likelihood weighting (generic sampling kernel), 10 k-node random DAG of arities 2-5, <= 3 parents, {int((ev >= 0).sum())} evidence nodes, "
            f"{samples} samples per call = per step" if generic else
            f"likelihood weighting, 10 k-node random DAG, {int((ev >= 0).sum())} evidence nodes, "
            f"{samples} samples per call = per step (BASELINE.json configs[4]: 10 M weighted samples)")
    out = {"workload": what, "value": rate, "unit": "samples/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
           "node_samples_per_s": rate * d.n, "roofline": roof}
    if not a.no_cpu:
        import oracle
        t0 = time.perf_counter()
        n_cpu = 4000
        oracle.lw_run(d, ev, n_cpu, seed=1)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n_cpu / dtc, "unit": "samples/s", "cores": 1, "kind": "port",
                               "sample": f"{n_cpu} samples of the same workload, oracle/lw_oracle.c, 1 thread"}
    if not generic and not getattr(a, "no_extras", False) and a.workload != "lw":
        try:
            out["generic_mixed10k"] = leg_lw(a, local_rank, torch, generic=True)
        except Exception as ex:  # noqa: BLE001
            out["generic_mixed10k"] = {"error": f"{type(ex).__name__}: {str(ex)[:300]}"}
    return out


def time_batches(eng, g, a, torch, sizes, cycled=True):
    """Throughput of bn_bp_run_batch_device over all sets of a call, for each batch size in `sizes`."""
    from bayesiannetwork_amd import synth
    out = {}
    for B in sizes:
        evs = [synth.random_evidence(g, a.evidence, seed=7 + q) for q in range(B)]
        eng.bp_set_evidence_batch(evs)
        for _ in range(3):
            r = eng.bp_run_batch_device(a.eps)
        torch.cuda.synchronize()
        steps = max(10, a.steps // 2)
        t0 = time.perf_counter()
        sweeps = 0
        for _ in range(steps):
            r = eng.bp_run_batch_device(a.eps)
            sweeps += int(r["sweeps"].sum())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = eng.bp_stats()
        out[f"B{B}"] = {"value": g.messages_per_sweep() * sweeps / dt, "unit": "edge-messages/s", "ms_per_call": dt / steps * 1e3,
                       "set_sweeps_per_call": sweeps / steps, "us_per_set_sweep": dt / sweeps * 1e6,
                       "algorithmic_gbs": st["algorithmic_bytes_per_sweep"] * sweeps / dt / 1e9, "path": PATH_NAME.get(eng.last_path()),
                       "what": "the SAME staged batch run again and again (a one-launch DAG batch then finds its evidence in place)"}
        if cycled:
            # two DIFFERENT batches alternate: every call has to put its evidence in force (a stream of different queries); staging
            # (bn_bp_set_evidence_batch: H2D of the packed sets) outside the clock, the clock around bn_bp_run_batch_device only
            evs2 = [synth.random_evidence(g, a.evidence, seed=1007 + q) for q in range(B)]
            for i in range(4):
                eng.bp_set_evidence_batch(evs2 if i & 1 else evs)
                eng.bp_run_batch_device(a.eps)
            dtc, swc = 0.0, 0
            for i in range(steps):
                eng.bp_set_evidence_batch(evs2 if i & 1 else evs)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = eng.bp_run_batch_device(a.eps)
                dtc += time.perf_counter() - t0
                swc += int(r["sweeps"].sum())
            out[f"B{B}"]["cycled"] = {"value": g.messages_per_sweep() * swc / dtc, "unit": "edge-messages/s", "ms_per_call": dtc / steps * 1e3,
                                     "set_sweeps_per_call": swc / steps, "us_per_set_sweep": dtc / swc * 1e6, "batches_cycled": 2,
                                     "what": "two different staged batches alternate (every call applies its evidence); staging outside the "
                                             "clock, the clock around bn_bp_run_batch_device only"}
    return out


def leg_batch(a, local_rank, torch):
    """Several evidence sets per call on the headline grid (bn_bp_run_batch_device: up to 4 sets walked round-robin
    by one resident launch, one CPT image in registers / LDS for all of them, each set's barrier hidden behind the
    others' sweeps; more sets = consecutive launches).  Throughput over all sets; every set's result is what a run
    of it alone gives (tests)."""
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import Engine
    g = synth.grid(a.rows, a.cols, 4, seed=2)
    out = {"workload": f"{a.rows}x{a.cols} grid, B evidence sets per call ({a.evidence:g} evidence each, different nodes), eps={a.eps:g}"}
    with Engine(g, device=local_rank) as eng:
        out.update(time_batches(eng, g, a, torch, (2, 4, 16)))
    return out


def leg_grid2048(a, local_rank, torch):
    """The HBM-resident point: 2048x2048 grid, 4.19 M nodes, 3.76 GB per sweep."""
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import Engine
    g = synth.grid(2048, 2048, 4, seed=2)
    ev = synth.random_evidence(g, a.evidence, seed=7)
    steps = 3
    with Engine(g, device=local_rank) as eng:
        eng.bp_set_evidence(ev)
        t = time_bp(eng, g, a.eps, steps, 1, torch, event_steps=2)
        cyc = time_cycled(eng, g, evidence_cycle(g, a.evidence, 4), a.eps, 4)
    out = {"workload": f"2048x2048 2D-grid BN, k=4, {g.n} nodes, {g.n_edges} edges, {ev.ne} evidence nodes, eps={a.eps:g} "
                       "(working set beyond the 256 MiB Infinity Cache)",
           "value": cyc["value"], "unit": "edge-messages/s", "ms_per_step": cyc["ms_per_step"], "steps": cyc["steps"],
           "sweeps_per_step": cyc["sweeps_per_step"], "messages_per_sweep": g.messages_per_sweep(),
           "cycled_evidence": cyc,
           "same_evidence": {"value": t["msgs"] / t["dt"], "ms_per_step": t["dt"] / steps * 1e3, "steps": steps,
                             "sweeps_per_step": t["sweeps_total"] / steps},
           "roofline": roofline_of(t, "grid2048")}
    if not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(g, ev, a.eps, budget_s=1.0, max_sweeps=3, max_runs=1)
    return out


def main():
    import faulthandler
    faulthandler.enable()   # a native crash prints the Python stack on stderr instead of dying silently
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["grid", "dag", "lw", "lwgen", "alarm", "mid", "batch", "dagbatch"], default="grid",
                    help="grid = BASELINE configs[2] (headline); dag = configs[1], 10 k-node random DAG; "
                         "lw = configs[4], likelihood weighting on the 10 k-node DAG")
    ap.add_argument("--samples", type=int, default=10000000, help="lw: weighted samples per step = per bn_lw_run call (BASELINE configs[4]: 10 M)")
    ap.add_argument("--rows", type=int, default=316)
    ap.add_argument("--cols", type=int, default=316)
    ap.add_argument("--eps", type=float, default=1e-3)
    ap.add_argument("--evidence", type=float, default=0.01)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-extras", dest="no_extras", action="store_true",
                    help="headline only: skip the config2_dag / config5_lw / grid2048 / host-to-host extras")
    ap.add_argument("--no-weak", dest="no_weak", action="store_true", help="N>1: skip the weak-scaling extra")
    ap.add_argument("--no-replicas", dest="no_replicas", action="store_true", help="N>1: skip the replicated-queries extra")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if "WORLD_SIZE" not in os.environ and a.gpus > 1:
            # `python bench.py --gpus N` typed plainly: this process -- which has made no GPU call, has not imported torch and has
            # not loaded the library -- starts the N ranks as fresh children under torch's launcher (the command the driver uses),
            # relays their output and exits with their code.  (Never exec or fork from a process that has initialised HIP.)
            raise SystemExit(benchline.launch_ranks(a.gpus, os.path.abspath(__file__), sys.argv[1:]))
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's world and --gpus must agree")

    import torch  # device plumbing only: barrier / synchronize around the timed region
    from bayesiannetwork_amd import synth
    from bayesiannetwork_amd.engine import Engine

    if world > 1 or os.environ.get("BN_FORCE_MULTI"):
        from bayesiannetwork_amd import multigpu
        if a.workload == "lw":
            return multigpu.bench_lw_main(a, rank, world, local_rank)
        return multigpu.bench_main(a, rank, world, local_rank)

    torch.cuda.set_device(local_rank)
    default_run = a.workload == "grid" and (a.rows, a.cols) == (316, 316)
    if a.workload in ("lw", "lwgen"):
        leg = leg_lw(a, local_rank, torch, generic=a.workload == "lwgen")
        out = {"metric": "weighted samples/sec (likelihood weighting)", "value": leg["value"], "unit": "samples/s",
               "n_gpus": 1, "steps": leg["steps"], "warmup": 2, "ms_per_step": leg["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": leg["workload"], "node_samples_per_s": leg["node_samples_per_s"]},
               "roofline": leg["roofline"]}
        if "cpu_baseline" in leg:
            out["cpu_baseline"] = leg["cpu_baseline"]
        benchline.emit(out)
        return
    if a.workload in ("batch", "dagbatch"):
        # 16 evidence sets per call (bn_bp_run_batch_device) on the headline grid / on BASELINE configs[1]: the legs `batch` and
        # `config2_dag.batch` of the default line on their own, for the profile passes (scripts/profile_bench.sh)
        g = synth.grid(a.rows, a.cols, 4, seed=2) if a.workload == "batch" else synth.random_dag(10000, 4, 64, 4, seed=1)
        with Engine(g, device=local_rank) as eng:
            b = time_batches(eng, g, a, torch, (16,))["B16"]
            st = eng.bp_stats()
        must_move = st["algorithmic_bytes_per_sweep"] - 8 * len(g.cpt)
        floor_us = must_move / (HBM_PEAK_GBS * 1e9) * 1e6
        out = {"metric": "edge-messages/sec to BP convergence, 16 evidence sets per call", "value": b["value"], "unit": "edge-messages/s", "n_gpus": 1,
               "steps": max(10, a.steps // 2), "warmup": 3, "ms_per_step": b["ms_per_call"], "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{g.name}: 16 evidence sets per call ({a.evidence:g} evidence each), eps={a.eps:g}", "run_path": b["path"],
                          "set_sweeps_per_call": b["set_sweeps_per_call"], "us_per_set_sweep": b["us_per_set_sweep"]},
               "roofline": {"bound": "hbm", "achieved": must_move / (b["us_per_set_sweep"] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": floor_us / b["us_per_set_sweep"], "traffic": None, "frac_resident": floor_us / b["us_per_set_sweep"],
                            "frac_survey_8d": b["algorithmic_gbs"] / HBM_PEAK_GBS, "must_move_bytes_per_set_sweep": must_move,
                            "floor_hbm_us": floor_us, "note": "frac = message + node-vector bytes of a set-sweep at the HBM peak / measured time per set-sweep "
                                                              "(the CPTs stay on chip and serve every set)"}}
        benchline.emit(out)
        return
    if a.workload == "alarm":
        leg = leg_alarm(a, local_rank, torch)
        out = {"metric": "queries/sec, evidence in -> marginals on the host (ALARM-shaped 37-node network)", "value": leg["value"],
               "unit": "queries/s", "n_gpus": 1, "steps": leg["host_to_host"]["steps"], "warmup": 2,
               "ms_per_step": leg["host_to_host"]["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic", "config": {"workload": leg["workload"], "run_path": leg["path"]},
               "roofline": {"bound": "latency", "note": "one workgroup on one CU: LDS round trips and chains of dependent fp64 additions in the "
                                                        "reference's order; neither HBM nor MFMA apply", "kernel_us_per_sweep": leg["kernel_us_per_sweep"]}}
        for k in ("tile_kernels", "batch", "cpu_baseline", "cpu_reference"):
            if k in leg:
                out[k] = leg[k]
        benchline.emit(out)
        return
    if a.workload == "mid":
        leg = leg_mid(a, local_rank, torch)
        out = {"metric": "queries/sec, evidence in -> marginals on the host (300-node mixed-arity network)", "value": 1e6 / leg["us_per_query"],
               "unit": "queries/s", "n_gpus": 1, "steps": 200, "warmup": 2, "ms_per_step": leg["us_per_query"] / 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": leg["workload"], "run_path": leg["path"]},
               "roofline": {"bound": "latency", "note": "a few tens of workgroups, a flag-per-workgroup grid barrier per phase; neither HBM nor MFMA apply",
                            "kernel_us_per_sweep": leg["kernel_us_per_sweep"]}}
        for k in ("workgroups", "tile_kernels", "batch_B64", "cpu_baseline", "mixed10k"):
            if k in leg:
                out[k] = leg[k]
        benchline.emit(out)
        return
    if a.workload == "dag":
        g = synth.random_dag(10000, 4, 64, 4, seed=1)
        wname = (f"10 k-node random DAG, <=4 parents, k=4, {g.n_edges} edges (BASELINE.json configs[1])")
        label = "dag10k"
    else:
        g = synth.grid(a.rows, a.cols, 4, seed=2)
        wname = (f"{a.rows}x{a.cols} 2D-grid BN, k=4, {g.n} nodes, {g.n_edges} edges (BASELINE.json configs[2])")
        label = {(316, 316): "grid316", (2048, 2048): "grid2048"}.get((a.rows, a.cols), "none")
    ev = synth.random_evidence(g, a.evidence, seed=7)
    eng = Engine(g, device=local_rank)
    eng.bp_set_evidence(ev)  # inputs resident in HBM before the timed region
    t = time_bp(eng, g, a.eps, a.steps, a.warmup, torch)
    if label == "grid316" and t["path"] == 0:
        label = "grid316_launch"  # the profile of the same grid with one launch per sweep (BN_MULTISWEEP=0)
    if label == "dag10k" and t["path"] == 0:
        label = "dag10k_launch"   # ... of the DAG with one launch per sweep (BN_DAG=0)
    roof = roofline_of(t, label)
    stream = measured_stream_gbs(torch, local_rank)
    # the yardstick beside the nominal 8 TB/s: the better of the library's copy and triad kernels; torch's copy_ beside it
    roof["hbm_stream_gbs_measured"] = max([v for v in (stream["copy"], stream["triad"]) if v] or [0.0]) or None
    roof["hbm_stream_gbs_copy"], roof["hbm_stream_gbs_triad"] = stream["copy"], stream["triad"]
    roof["hbm_stream_gbs_torch_copy"] = stream["torch_copy"]
    if roof["hbm_stream_gbs_measured"] and roof.get("traffic_gbs"):
        roof["frac_of_measured_stream"] = roof["traffic_gbs"] / roof["hbm_stream_gbs_measured"]   # counter traffic per second over what a pure stream reaches
    out = {
        "metric": "edge-messages/sec to BP convergence", "value": t["msgs"] / t["dt"], "unit": "edge-messages/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": t["dt"] / a.steps * 1e3,
        # the N-GPU lines cut this same network into N stripes (BASELINE configs[3]): total work is fixed
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{wname}, {ev.ne} evidence nodes, eps={a.eps:g}",
                   "sweeps_per_step": t["sweeps_total"] / a.steps, "messages_per_sweep": g.messages_per_sweep(),
                   "parallelism": "1 GPU", "run_path": PATH_NAME.get(t["path"], "?")},
        "roofline": roof,
        "sweep_only_msgs_per_s": g.messages_per_sweep() / t["avg_sweep_s"],
    }
    if not a.no_extras:
        h2h = time_host_to_host(eng, g, evidence_cycle(g, a.evidence), a.eps, max(min(a.steps, 40), 16))
        out["cycled_evidence"] = time_cycled(eng, g, evidence_cycle(g, a.evidence), a.eps, max(min(a.steps, 40), 16))
        out["value_host_to_host"] = h2h["value"]
        # SURVEY 8(d)'s own definition of the roofline fraction: algorithmic bytes per edge-message x messages/s over the HBM peak
        bytes_per_msg = t["stats"]["algorithmic_bytes_per_sweep"] / g.messages_per_sweep()
        out["frac_host_to_host"] = h2h["value"] * bytes_per_msg / (HBM_PEAK_GBS * 1e9)
        out["ms_per_step_host_to_host"] = h2h["ms_per_step"]
        out["host_to_host"] = h2h
        # the same scalars inside `config` (a reader that keeps only the contract's keys still sees them)
        out["config"].update({"value_host_to_host": h2h["value"], "frac_host_to_host_survey_8d": out["frac_host_to_host"],
                              "ms_per_step_host_to_host": h2h["ms_per_step"]})
    if not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(g, ev, a.eps)
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        if ncpu > 1:
            out["cpu_baseline_all_cores"] = cpu_baseline(g, ev, a.eps, budget_s=8.0, threads=min(ncpu, 64))
        ref = cpu_reference_small()
        if ref:
            out["cpu_reference_small"] = ref
    eng.close()
    if default_run and not a.no_extras:
        try:
            dc = leg_dropin_cpp(a, local_rank, torch)
        except Exception as ex:  # noqa: BLE001 - an extra must never lose the headline
            dc = {"error": f"{type(ex).__name__}: {str(ex)[:300]}"}
        out["dropin_cpp"] = dc
        if "config3_grid" in dc:
            gd = dc["config3_grid"]
            out["config"].update({"ms_per_query_dropin_cpp": gd["operator_ms"], "ms_per_query_dropin_cpp_run_view": gd["run_view_ms"],
                                  "ms_per_query_dropin_cpp_run_prepared": gd.get("run_prepared_ms"),
                                  "ms_dropin_cpp_map_build": gd["map_build_ms"], "ms_dropin_cpp_map_destroy": gd["map_destroy_ms"],
                                  "dropin_cpp_matches_c_abi": gd.get("matches_c_abi")})
            # SURVEY 8(d)'s `t` (evidence in -> marginals on the host) as a COMPILED caller of the C ABI sees it: the same bn_bp_run_view
            # call, timed in C++ (the host_to_host leg above goes through Python / ctypes: ~25 us more per query)
            if gd.get("c_abi_ms"):
                v_cpp = g.messages_per_sweep() * gd["sweeps_per_query"] / (gd["c_abi_ms"] * 1e-3)
                out["config"].update({"ms_per_step_host_to_host_cpp": gd["c_abi_ms"], "value_host_to_host_cpp": v_cpp,
                                      "frac_host_to_host_cpp_survey_8d": v_cpp * t["stats"]["algorithmic_bytes_per_sweep"] / g.messages_per_sweep() / (HBM_PEAK_GBS * 1e9)})
        for key, fn in (("batch", leg_batch), ("config1_alarm", leg_alarm), ("mid_mixed300", leg_mid), ("config2_dag", leg_dag), ("config5_lw", leg_lw), ("grid2048", leg_grid2048)):
            try:
                out[key] = fn(a, local_rank, torch)
            except Exception as ex:  # noqa: BLE001 - an extra must never lose the headline
                out[key] = {"error": f"{type(ex).__name__}: {str(ex)[:300]}"}
    benchline.emit(out)


if __name__ == "__main__":
    main()

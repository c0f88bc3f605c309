"""bench.py's output contract and its N-rank launcher (bayesiannetwork_amd/benchline.py), on the CPU.

Round 5's bench printed ONE line of 21.5 KB and the driver could not parse it: the last stdout line now
carries the contract only, at most 4 KB, and these tests hold it there -- fed with the round-5 record itself."""
import io
import json
import os
import subprocess
import sys
import textwrap

import pytest

from bayesiannetwork_amd import benchline
from helpers import parse_bench_output

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _canned():
    """The full record of the round-5 default run (ten workloads, 21.5 KB as one line)."""
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))


def test_contract_line_fits_and_keeps_the_contract():
    full = _canned()
    assert len(json.dumps(full)) > 20000   # (the record that broke the driver's reader)
    s = benchline.contract_line(full)
    assert len(s.encode()) <= benchline.MAX_LINE_BYTES == 4096
    assert "\n" not in s
    line = json.loads(s)
    # the contract's keys, in the contract's order, and nothing else
    assert list(line.keys()) == [k for k in benchline.CONTRACT_KEYS if k in full]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-6) and line["dtype"] == "f64"
    assert "configs[2]" in line["config"]["workload"]
    # roofline: scalars only, the five of the contract first, frac beside SURVEY 8(d)'s figure
    roof = line["roofline"]
    assert list(roof.keys())[:6] == ["bound", "achieved", "peak", "unit", "frac", "traffic"]
    assert all(not isinstance(v, (dict, list)) for v in roof.values())
    assert "note" not in roof and "limiter" not in roof
    assert roof["frac"] == pytest.approx(full["roofline"]["frac_resident"], rel=1e-6)
    assert roof["frac_survey_8d"] == pytest.approx(full["roofline"]["frac_survey_8d"], rel=1e-6)
    for k in ("kernel", "avg_launch_us", "sweeps_per_launch", "avg_sweep_us"):
        assert k in roof
    cpu = line["cpu_baseline"]
    assert set(cpu) == {"value", "unit", "cores", "kind", "sample"} and cpu["kind"] == "port" and cpu["cores"] == 1
    # config: the host-to-host and class-surface scalars a reader of the contract keys should see
    for k in ("value_host_to_host", "frac_host_to_host_survey_8d", "ms_per_step_host_to_host", "ms_per_query_dropin_cpp"):
        assert k in line["config"]
    assert all(not isinstance(v, (dict, list)) for v in line["config"].values())


def test_emit_prints_the_extras_first_and_the_contract_line_last():
    full = _canned()
    buf = io.StringIO()
    s = benchline.emit(full, stream=buf, side_file=False)
    text = buf.getvalue()
    assert text.endswith(s + "\n")
    line, extras = parse_bench_output(text)
    assert line == json.loads(s)
    # every workload of the full record is still in the output, each as a line of its own with its own roofline / cpu_baseline
    for k in ("batch", "config1_alarm", "mid_mixed300", "config2_dag", "config5_lw", "grid2048", "dropin_cpp", "host_to_host",
              "cycled_evidence", "cpu_baseline_all_cores"):
        assert k in extras, k
    assert "roofline" in extras["config2_dag"] and "cpu_baseline" in extras["config2_dag"]
    assert extras["roofline_full"]["resident"]["frac_resident"] == full["roofline"]["frac_resident"]
    assert extras["scalars"]["value_host_to_host"] == full["value_host_to_host"]
    # a reader that keeps only the last 8 KB of stdout still gets the whole contract line
    tail = text[-8192:]
    assert json.loads(tail.splitlines()[-1]) == line


def test_a_line_that_cannot_fit_sheds_optional_keys_then_fails_loudly():
    full = _canned()
    full["config"] = dict(full["config"], run_path="x" * 3000, parallelism="y" * 3000, **{f"pad{i}": "z" * 300 for i in range(12)})
    line = json.loads(benchline.contract_line(full))   # (strings are cut to 400 characters, then whole keys go, these two first)
    assert "parallelism" not in line["config"] and "run_path" not in line["config"] and "workload" in line["config"]
    assert "value_host_to_host" in line["config"]      # the short scalars outlive the padding
    full["config"]["workload"] = "w" * 5000   # the workload is never dropped (truncated to 400), so this still fits
    assert len(benchline.contract_line(full).encode()) <= 4096
    full["metric"] = "m" * 5000
    with pytest.raises(ValueError):
        benchline.contract_line(full)


def _stub(tmp_path, body):
    p = tmp_path / "stub_ranks.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_launch_ranks_relays_and_ends_with_rank0s_line(tmp_path):
    """`python bench.py --gpus N` typed plainly: the parent relays the children's output; the contract line is the last line of
    ITS output even when a rank printed something after it."""
    contract = json.dumps({"metric": "edge-messages/sec to BP convergence", "value": 1.5, "n_gpus": 2})
    cmd = _stub(tmp_path, f"""
        import os, sys
        assert "WORLD_SIZE" not in os.environ   # (the real command is torch's launcher, which sets it for the ranks)
        print('{{"extra": "weak_scaling", "record": {{"value": 2.0}}}}')
        print('{contract}')
        print("[rank 1] late chatter", flush=True)
    """)
    buf = io.StringIO()
    rc = benchline.launch_ranks(2, "bench.py", ["--gpus", "2"], cmd=cmd, stream=buf)
    assert rc == 0
    line, extras = parse_bench_output(buf.getvalue())
    assert line["n_gpus"] == 2 and extras["weak_scaling"]["value"] == 2.0
    assert buf.getvalue().splitlines().count(contract) == 2   # relayed where it came, repeated as the last line


def test_launch_ranks_reports_failures(tmp_path):
    buf = io.StringIO()
    assert benchline.launch_ranks(2, "bench.py", [], cmd=_stub(tmp_path, "import sys; print('boom'); sys.exit(3)"), stream=buf) == 3
    # a job that prints no result line is a failure even with exit code 0
    assert benchline.launch_ranks(2, "bench.py", [], cmd=_stub(tmp_path, "print('nothing useful')"), stream=buf) == 1
    # the watchdog ends a job that hangs (the process group this call created)
    assert benchline.launch_ranks(2, "bench.py", [], cmd=_stub(tmp_path, "import time; time.sleep(60)"), timeout_s=1.0, stream=buf) == 124


def test_rank_command_is_the_drivers_launch_line():
    cmd = benchline.rank_command(8, "/x/bench.py", ["--gpus", "8", "--steps", "20"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "20"]


def test_bench_gpus_n_without_a_launcher_starts_one(tmp_path):
    """bench.py itself: --gpus 2 and no WORLD_SIZE -> the parent neither imports torch nor loads the library, it hands over to
    launch_ranks (here with a stub for the launcher's command line) and exits with its code; --gpus N under a launcher whose world
    differs is still refused."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    probe = textwrap.dedent(f"""
        import json, sys
        sys.argv = ["bench.py", "--gpus", "2", "--steps", "3"]
        sys.path.insert(0, {ROOT!r})
        import bench
        from bayesiannetwork_amd import benchline
        seen = {{}}
        def fake(n, script, argv, **kw):
            seen.update(n=n, script=script, argv=argv, torch="torch" in sys.modules, lib=benchline.__name__ and "bayesiannetwork_amd._lib" in sys.modules)
            return 7
        benchline.launch_ranks = fake
        try:
            bench.main()
        except SystemExit as ex:
            seen["rc"] = ex.code
        print(json.dumps(seen))
    """)
    p = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    seen = json.loads(p.stdout.strip().splitlines()[-1])
    assert seen["n"] == 2 and seen["rc"] == 7 and seen["argv"] == ["--gpus", "2", "--steps", "3"]
    assert seen["script"].endswith("bench.py") and seen["torch"] is False and seen["lib"] is False
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert p.returncode != 0 and "must agree" in p.stderr


def test_design_figures_script_runs_on_the_committed_profiles():
    """scripts/design_figures.py (the figures DESIGN.md quotes) reads profiles/r06_summary.json + r06_bench_line.json: it runs, names the
    profiled library, and prices the headline kernel and the grid batch the way DESIGN section 4.2 states them."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "design_figures.py"), "r06"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    out = p.stdout
    summary = json.load(open(os.path.join(ROOT, "profiles", "r06_summary.json")))
    assert summary["failed_passes"] == [] and summary["lib_sha256"][:16] in out
    assert "bp_resident_kernel" in out and "bp_dag_kernel" in out and "grid batch:" in out and "configs[1]:" in out
    batch = next(ln for ln in out.splitlines() if ln.startswith("batch_grid316"))
    assert "per set-sweep" in batch and "of the measured stream" in batch and "x must-move" in batch
    # the full record of the round's default run is itself a valid input of the line builder
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_line.json")))
    line = json.loads(benchline.contract_line(full))
    assert line["roofline"]["hbm_stream_gbs_measured"] > 5500 and line["roofline"]["traffic"] > 0 and "configs[2]" in line["config"]["workload"]
    for k in ("batch", "config2_dag", "config5_lw", "grid2048", "config1_alarm", "mid_mixed300", "dropin_cpp"):
        assert k in full and "error" not in full[k], k
    assert full["batch"]["B16"]["cycled"]["value"] > 3e10 and full["config5_lw"]["generic_mixed10k"]["value"] > 1e7

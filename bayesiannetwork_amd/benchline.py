"""What bench.py prints, and how `bench.py --gpus N` starts its ranks.

The driver reads the LAST stdout line of a bench run and keeps only a few KB of tail beside it: that line carries the
contract and nothing else (`contract_line`, <= MAX_LINE_BYTES); every other record of a run -- the extra workloads, the
full roofline record with its counters and provenance -- goes out BEFORE it, one JSON object per line
(`{"extra": name, "record": {...}}`), and into gpurun_out/bench_full_n<N>.json.

Nothing in this module imports torch or loads the library: `launch_ranks` runs in a parent process that must stay
free of any GPU state (a process that has initialised HIP is never replaced or forked into ranks)."""
from __future__ import annotations

import json
import os
import signal
import socket
import subprocess
import sys
import threading

MAX_LINE_BYTES = 4096

# the contract's keys, in the order the line carries them
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
# roofline: scalars only, the contract's five first.  frac = the bound that applies to the dominant kernel (frac_resident on
# the one-launch paths that keep the CPTs on chip), frac_survey_8d = SURVEY 8(d)'s algorithmic bytes / time / peak beside it.
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_survey_8d", "achieved_survey_8d", "frac_resident",
                 "kernel", "avg_launch_us", "sweeps_per_launch", "avg_sweep_us", "avg_sweep_us_devclock",
                 "algorithmic_bytes_per_launch", "must_move_bytes_per_sweep", "floor_hbm_us", "floor_valu_us", "traffic_gbs",
                 "traffic_stale", "valu_frac", "hbm_stream_gbs_measured", "hbm_stream_gbs_torch_copy", "frac_of_measured_stream")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")
# dropped from `config` first when a line would not fit (never the workload)
CONFIG_OPTIONAL_LAST = ("run_path", "parallelism")


def _scalar(v):
    return v is None or isinstance(v, (bool, int, float, str))


def _round(v, digits=7):
    """Floats to `digits` significant figures: the line is read by people and a size-limited reader."""
    if isinstance(v, float) and v == v and v not in (float("inf"), float("-inf")):
        return float(f"{v:.{digits}g}")
    return v


def _compact(d, keys=None, max_str=400):
    out = {}
    for k in (keys if keys is not None else d.keys()):
        if k in d and _scalar(d[k]):
            v = _round(d[k])
            if isinstance(v, str) and len(v) > max_str:
                v = v[:max_str - 3] + "..."
            out[k] = v
    return out


def contract_line(out: dict) -> str:
    """The one line the driver parses: the contract's keys only, `config` / `roofline` / `cpu_baseline` flattened to
    scalars (nested records, counter dumps and prose stay in the extras), at most MAX_LINE_BYTES bytes."""
    line = {}
    for k in CONTRACT_KEYS:
        if k not in out:
            continue
        v = out[k]
        if k == "config":
            v = _compact(v)
        elif k == "roofline":
            v = _compact(v, ROOFLINE_KEYS, max_str=120)
        elif k == "cpu_baseline":
            v = _compact(v, CPU_KEYS, max_str=200)
        else:
            v = _round(v)
        line[k] = v
    s = json.dumps(line)
    # a line that still does not fit sheds the optional parts, longest strings of `config` first
    cfg = line.get("config", {})
    for k in CONFIG_OPTIONAL_LAST + tuple(sorted((k for k in cfg if k != "workload"), key=lambda k: -len(json.dumps(cfg[k])))):
        if len(s.encode()) <= MAX_LINE_BYTES:
            break
        cfg.pop(k, None)
        s = json.dumps(line)
    if len(s.encode()) > MAX_LINE_BYTES:
        roof = line.get("roofline", {})
        for k in ROOFLINE_KEYS[::-1]:
            if len(s.encode()) <= MAX_LINE_BYTES or k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
                break
            roof.pop(k, None)
            s = json.dumps(line)
    if len(s.encode()) > MAX_LINE_BYTES:
        raise ValueError(f"bench line is {len(s.encode())} bytes (> {MAX_LINE_BYTES})")
    return s


def extras_of(out: dict) -> dict:
    """Everything of a full record the contract line does not carry: the extra workloads, and the full `roofline` / `config` /
    `cpu_baseline` records when they hold more than the line shows."""
    ex = {k: v for k, v in out.items() if k not in CONTRACT_KEYS and not _scalar(v)}
    scalars = {k: v for k, v in out.items() if k not in CONTRACT_KEYS and _scalar(v)}
    if scalars:
        ex = dict({"scalars": scalars}, **ex)
    for k, keys in (("roofline", ROOFLINE_KEYS), ("config", None), ("cpu_baseline", CPU_KEYS)):
        v = out.get(k)
        if isinstance(v, dict) and any(not _scalar(x) or (keys is not None and kk not in keys) for kk, x in v.items()):
            ex[k + "_full"] = v
    return ex


def emit(out: dict, stream=None, side_file: bool = True) -> str:
    """Print a bench record: extras first (one JSON object per line), the contract line LAST.  Returns the contract line."""
    stream = stream or sys.stdout
    line = contract_line(out)   # (before anything is printed: a record that cannot make a line fails loudly)
    for name, rec in extras_of(out).items():
        print(json.dumps({"extra": name, "record": rec}), file=stream)
    if side_file:
        try:   # best effort: scratch on the GPU box, merged back by gpurun
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            d = os.path.join(root, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, f"bench_full_n{out.get('n_gpus', 1)}.json"), "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            pass
    if stream is sys.stdout:
        try:   # native libraries of this process (gloo, RCCL) write through C stdio: what they have buffered goes out BEFORE the line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
    print(line, file=stream, flush=True)
    return line


# ---- bench.py --gpus N typed plainly: the parent starts the ranks -------------------------------------------------------

def free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def rank_command(n: int, script: str, argv: list, port: int) -> list:
    """One process per GPU under torch's launcher, the way the driver itself starts an N-GPU bench."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script] + list(argv)


def launch_ranks(n: int, script: str, argv: list, cmd: list | None = None, timeout_s: float | None = None, stream=None) -> int:
    """`python bench.py --gpus N` without WORLD_SIZE in the environment: start the N ranks as FRESH child processes (this parent
    has made no GPU call and makes none), relay their stdout line by line, print rank 0's contract line again as the last
    line if anything followed it, and return non-zero if the launcher did.  A watchdog ends the child's process group -- the
    one this function created -- when the job exceeds `timeout_s`."""
    stream = stream or sys.stdout
    if timeout_s is None:
        timeout_s = float(os.environ.get("BN_BENCH_WATCHDOG_S", "600")) + 120.0
    cmd = cmd or rank_command(n, script, argv, free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, start_new_session=True)
    timed_out = []

    def _expire():
        timed_out.append(True)
        try:
            os.killpg(p.pid, signal.SIGKILL)   # the session this call started: its pgid is the child's pid
        except OSError:
            pass

    dog = threading.Timer(timeout_s, _expire)
    dog.daemon = True
    dog.start()
    last_contract, last_line = None, None
    try:
        for ln in p.stdout:
            ln = ln.rstrip("\n")
            print(ln, file=stream, flush=True)
            last_line = ln
            if ln.startswith("{") and '"metric"' in ln:
                try:
                    if "metric" in json.loads(ln):
                        last_contract = ln
                except ValueError:
                    pass
        rc = p.wait()
    finally:
        dog.cancel()
    if timed_out:
        print(f"[bench] the {n}-rank job exceeded {timeout_s:.0f} s and was ended", file=sys.stderr, flush=True)
        return 124
    if rc == 0 and last_contract is None:
        print(f"[bench] the {n}-rank job printed no result line", file=sys.stderr, flush=True)
        return 1
    if rc == 0 and last_line != last_contract:
        print(last_contract, file=stream, flush=True)
    return rc

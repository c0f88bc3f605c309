# Where an any-arity tile spends its time: needs the diagnostic library
#   (cd build/dbg_csrc && make EXTRA=-DBN_TILE_CLOCK OUT=../libbn_dbg.so)   and   BN_MI355X_LIB=build/libbn_dbg.so
# Prints, per tile of the last sweep launch, the 100 MHz clock differences between the stamps in tile_flat.
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import Evidence, _lib, synth  # noqa: E402
from bayesiannetwork_amd.dsc import load_dsc  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "alarm_shaped"
if name == "alarm_shaped":
    m, _ = load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))
elif name == "mixed300":
    m = synth.random_dag(300, 3, 32, [2, 3, 4, 3, 2, 5], seed=4)
elif name.startswith("grid"):
    m = synth.grid(int(name[4:]), int(name[4:]), 4, seed=2)
elif name == "dag10k":
    m = synth.random_dag(10000, 4, 64, 4, seed=1)
elif name == "dag200":
    m = synth.random_dag(200, 4, 64, 4, seed=200)
elif name == "mixed2k":
    m = synth.random_dag(2000, 4, 64, [2, 3, 4, 3, 2, 4, 4], seed=9)
else:
    m = synth.random_dag(60, 3, 16, [2, 3, 4, 3, 2, 4, 4], seed=9)
L = _lib.lib()
with Engine(m) as e:
    e.set_option("multisweep", 0)
    e.bp_set_evidence(Evidence.none())
    for _ in range(4):
        r = e.bp_run_device(1e-6)
    n = 4096  # stamps exist for the first 4096 lane slots only
    buf = np.zeros((n, 12), dtype=np.uint64)
    rows = []
    for getter in ("bn_debug_tile_clock", "bn_debug_tile_clock_light", "bn_debug_tile_clock_ug", "bn_debug_tile_clock_u"):
        # every translation unit with sweep kernels keeps its own stamps; the one that ran has non-zero ones
        rc = getattr(L, getter)(buf.ctypes.data_as(ctypes.c_void_p), n)
        assert rc == 0, rc
        rows = [(i, b.copy()) for i, b in enumerate(buf) if b[11] != 0]
        if rows:
            break
    t0 = min(int(b[9]) for _, b in rows)
    print(name, "sweeps", r["sweeps"], "tiles stamped", len(rows), "(x10 ns; columns: entry->s0 desc, s1 class, s2 loads issued, s3 inputs arrived,"
          " s4 pi summed, s5 lambda summed, s6 normalised+stored, s7 children staged, s8 parent role, end)")
    for i, b in rows:
        d = int(b[10])
        if d == 0:
            print(f"slot {i:4d} (not an any-arity tile) start {int(b[9]) - t0:5d} total {int(b[11]) - int(b[9]):5d}")
            continue
        G, mm, kv, cmax, nrows = d & 255, (d >> 8) & 255, (d >> 16) & 255, (d >> 24) & 255, d >> 32
        nst = 7 if nrows == 0xffff else (6 if nrows == 0xfffe else 9)  # lane-group tiles stamp 7 points, one-lane tiles 6 (G column = RC), any-arity tiles 9
        st = [int(b[9])] + [int(b[k]) for k in range(nst)] + [int(b[11])]
        diffs = [st[k + 1] - st[k] for k in range(len(st) - 1)]
        print(f"slot {i:4d} G={G:2d} m={mm} kv={kv} cmax={cmax} rows={nrows:4d} start {st[0] - t0:5d} total {st[-1] - st[0]:5d} :", " ".join(f"{x:4d}" for x in diffs))

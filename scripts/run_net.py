# Runs belief propagation on one named small network a few times (a target for rocprofv3 passes):
#   python3 scripts/run_net.py alarm_shaped|mixed300|dag200|grid32 [reps]
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesiannetwork_amd import Evidence, synth  # noqa: E402
from bayesiannetwork_amd.dsc import load_dsc  # noqa: E402
from bayesiannetwork_amd.engine import Engine  # noqa: E402

name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if name == "alarm_shaped":
    m, _ = load_dsc(os.path.join(ROOT, "tests", "golden", "alarm_shaped.dsc"))
elif name == "mixed300":
    m = synth.random_dag(300, 3, 32, [2, 3, 4, 3, 2, 5], seed=4)
elif name == "mixed2k":
    m = synth.random_dag(2000, 4, 64, [2, 3, 4, 3, 2, 4, 4], seed=9)
elif name == "dag200":
    m = synth.random_dag(200, 4, 64, 4, seed=200)
else:
    m = synth.grid(32, 32, 4, seed=1)
with Engine(m) as e:
    e.set_option("multisweep", 0)
    e.bp_set_evidence(Evidence.none())
    for _ in range(reps):
        r = e.bp_run_device(1e-6)
    print(name, r["sweeps"], e.layout()["n_tiles"])

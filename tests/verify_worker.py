"""Worker of tests/test_dist_cpu.py::test_verify_in_kernel_exchange_keeps_one_collective_schedule (gloo, CPU).
multigpu.verify_in_kernel_exchange with stand-ins for the shard engine and the unsharded reference engine: on the rank named on the
command line the `fail_at`-th verification run raises (a bounded wait gave up) or returns different bits -- while the peer's same run
succeeds.  Every rank must leave the verification together with the same answer (a rank that jumped from a run straight
to the final all-reduce while its peer went on to the next barrier paired a barrier with an all-reduce -- a hang on gloo)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bayesiannetwork_amd import multigpu, synth  # noqa: E402

MODEL = synth.grid(4, 4, 4, seed=1)
NBEL = int(MODEL.k.sum())


class RefEngine:   # stands in for the unsharded Engine(model, device) the verification compares with
    def __init__(self, model, device=0):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        pass

    def bp_run(self, ev, eps):
        return {"sweeps": 5, "beliefs": np.full(NBEL, 0.25)}

    def bp_residuals(self):
        return np.array([0.5, 0.1, 0.01, 0.001, 0.0001])


class ShardStub:
    def __init__(self, rank, mode, fail_at):
        self.rank, self.mode, self.fail_at = rank, mode, fail_at
        self.runs, self.options, self.staged = 0, {}, 0

    def info(self, name):
        assert name == "shard_flow"
        return 1

    def node_slots(self):
        return np.arange(MODEL.n)

    def bp_set_evidence(self, ev):
        self.staged += 1

    def bp_run_device(self, eps):
        self.runs += 1
        if self.mode == "raise" and self.runs == self.fail_at:
            raise RuntimeError("in-kernel exchange gave up a bounded wait")
        return {"sweeps": 5}

    def bp_beliefs(self):
        bel = np.full(NBEL, 0.25)
        if self.mode == "bits" and self.runs == self.fail_at:
            bel[3] = 0.2500000001
        return bel

    def bp_residuals(self):
        return np.array([0.5, 0.1, 0.01, 0.001, 0.0001])

    def last_path(self):
        return 2

    def set_option(self, name, value):
        self.options[name] = value


def main():
    failing, mode, fail_at = sys.argv[1], sys.argv[2], int(sys.argv[3])
    multigpu.Engine = RefEngine
    dist = multigpu.init_control_plane()
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = failing != "none" and int(failing) == rank
    eng = ShardStub(rank, mode if mine else "ok", fail_at)
    ok = multigpu.verify_in_kernel_exchange(eng, MODEL, synth.random_evidence(MODEL, 0.1, seed=2), 1e-3, device=0, repeats=3)
    if failing == "none":
        assert ok and eng.runs == 12 and "multisweep" not in eng.options          # 4 evidence sets x 3 runs
    else:   # every rank says no, switches to the RCCL exchange, and stopped after the SAME run
        assert not ok and eng.options.get("multisweep") == 0
        assert eng.runs == fail_at, (rank, eng.runs, fail_at)
    dist.barrier()   # (would hang here, or before, if the ranks had issued different collectives)
    if rank == 0:
        print(f"VERIFY_OK failing={failing} mode={mode} world={world}", flush=True)


if __name__ == "__main__":
    main()

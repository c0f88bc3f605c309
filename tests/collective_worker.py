"""Worker of tests/test_dist_cpu.py::test_run_collective_resolves_a_failed_in_kernel_run_on_every_rank (gloo, CPU).
multigpu.run_collective with a stand-in for the shard engine: the rank named on the command line gets BN_ERR_STATE from its first
run (what the library returns when the in-kernel exchange gives up a bounded wait) -- or no rank does ("none")."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bayesiannetwork_amd import _lib, multigpu  # noqa: E402


class StubEngine:
    def __init__(self, rank, fails):
        self.rank, self.fails = rank, fails
        self.multisweep, self.calls = 1, []

    def bp_run_device(self, eps, max_sweeps=0):
        self.calls.append(self.multisweep)
        if self.fails and self.multisweep == 1:
            raise _lib.BnError(_lib.BN_ERR_STATE, "in-kernel exchange gave up a bounded wait")
        return {"sweeps": 7, "path": 2 if self.multisweep else 0}

    def set_option(self, name, value):
        assert name == "multisweep"
        self.multisweep = value


def main():
    failing = sys.argv[1]
    dist = multigpu.init_control_plane()
    rank, world = dist.get_rank(), dist.get_world_size()
    eng = StubEngine(rank, failing != "none" and int(failing) == rank)
    res, still = multigpu.run_collective(eng, 1e-3)
    if failing == "none":
        assert still and eng.calls == [1] and res["path"] == 2
    else:   # EVERY rank repeated the run on the RCCL exchange, also the ones whose first run had succeeded
        assert not still and eng.calls == [1, 0] and eng.multisweep == 0 and res["path"] == 0
    # an error that is not "the exchange gave up" is not swallowed (every rank raises it here: a one-sided raise would leave the
    # others in the all-reduce, which is the caller's problem exactly as with any other collective)
    class Broken(StubEngine):
        def bp_run_device(self, eps, max_sweeps=0):
            raise _lib.BnError(_lib.BN_ERR_ARG, "bad argument")
    try:
        multigpu.run_collective(Broken(rank, False), 1e-3)
        raise SystemExit("BN_ERR_ARG was swallowed")
    except _lib.BnError as ex:
        assert ex.code == _lib.BN_ERR_ARG
    dist.barrier()
    if rank == 0:
        print(f"COLLECTIVE_OK failing={failing} world={world}", flush=True)


if __name__ == "__main__":
    main()

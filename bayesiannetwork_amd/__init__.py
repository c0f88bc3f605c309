"""bayesiannetwork_amd -- MI355X-native Loopy Belief Propagation / Likelihood Weighting.

Host-side mirror of the reference's ``bn::inference`` functors over hand-written HIP kernels
(``csrc/``) behind the C ABI declared in ``include/bn_mi355x.h``.  The HIP library is loaded on
first use and its absence is a hard error: there is no CPU fallback in this package.
"""
from .flat import Evidence, FlatModel, from_parent_lists  # noqa: F401
from . import synth  # noqa: F401

__all__ = ["Evidence", "FlatModel", "from_parent_lists", "synth"]

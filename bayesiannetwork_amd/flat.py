"""Flat Bayesian-network model: the host-side data format that crosses the C ABI.

This is what ``bn::graph_t`` + ``bn::cpt_t`` (reference ``bayesian/graph.hpp:57-161,173-486``)
flatten to.  Node identity is the position in ``graph_t::vertex_list()`` (``graph.hpp:214``),
never ``vertex_t::id``.  Parents are listed in ascending node id, the order
``graph_t::in_edges`` produces (``graph.hpp:389-402``).  The CPT of a node is row-major:
row = mixed radix over the parents with the FIRST parent most significant (the enumeration
order of ``all_combination_pattern``, ``belief_propagation.hpp:269-295``), own state fastest.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class FlatModel:
    k: np.ndarray          # int32 [n]      vertex_t::selectable_num
    in_ptr: np.ndarray     # int32 [n+1]    CSR row pointers over parents
    in_idx: np.ndarray     # int32 [E]      parents, ascending per node
    cpt_off: np.ndarray    # int64 [n+1]    prefix sums of CPT sizes
    cpt: np.ndarray        # float64        flat CPTs
    name: str = ""
    meta: dict = field(default_factory=dict)

    def __post_init__(self):
        self.k = np.ascontiguousarray(self.k, dtype=np.int32)
        self.in_ptr = np.ascontiguousarray(self.in_ptr, dtype=np.int32)
        self.in_idx = np.ascontiguousarray(self.in_idx, dtype=np.int32)
        self.cpt_off = np.ascontiguousarray(self.cpt_off, dtype=np.int64)
        self.cpt = np.ascontiguousarray(self.cpt, dtype=np.float64)

    # ---- sizes -----------------------------------------------------------------
    @property
    def n(self) -> int:
        return int(self.k.shape[0])

    @property
    def n_edges(self) -> int:
        return int(self.in_ptr[-1]) if self.in_ptr.size else 0

    @property
    def node_off(self) -> np.ndarray:
        """Offsets of each node's k-vector in node-major belief / histogram arrays."""
        off = np.zeros(self.n + 1, dtype=np.int64)
        np.cumsum(self.k, out=off[1:])
        return off

    @property
    def msg_off(self) -> np.ndarray:
        """Offsets of each CSR edge's message (length k[parent]) in edge-major arrays."""
        off = np.zeros(self.n_edges + 1, dtype=np.int64)
        if self.n_edges:
            np.cumsum(self.k[self.in_idx], out=off[1:])
        return off

    def parents(self, v: int) -> np.ndarray:
        return self.in_idx[self.in_ptr[v]:self.in_ptr[v + 1]]

    def cpt_of(self, v: int) -> np.ndarray:
        kv = int(self.k[v])
        return self.cpt[self.cpt_off[v]:self.cpt_off[v + 1]].reshape(-1, kv)

    # ---- checks ----------------------------------------------------------------
    def validate(self) -> None:
        """Raise ValueError on anything the C ABI would reject."""
        n = self.n
        if self.in_ptr.shape[0] != n + 1 or self.cpt_off.shape[0] != n + 1:
            raise ValueError("in_ptr / cpt_off must have n+1 entries")
        if n and (self.k <= 0).any():
            raise ValueError("every node needs selectable_num >= 1")
        if self.in_ptr[0] != 0 or (np.diff(self.in_ptr) < 0).any():
            raise ValueError("in_ptr must be a non-decreasing prefix array starting at 0")
        if self.in_idx.shape[0] != self.n_edges:
            raise ValueError("in_idx length != in_ptr[n]")
        if self.n_edges and ((self.in_idx < 0) | (self.in_idx >= n)).any():
            raise ValueError("parent index out of range")
        if self.n_edges:
            child = np.repeat(np.arange(n, dtype=np.int32), np.diff(self.in_ptr))
            if (self.in_idx == child).any():
                raise ValueError("a node is its own parent")
            same = child[1:] == child[:-1]
            if (same & (np.diff(self.in_idx) <= 0)).any():
                raise ValueError("parents of a node must be strictly ascending")
        sizes = self.expected_cpt_sizes()
        if (np.diff(self.cpt_off) != sizes).any() or self.cpt_off[0] != 0:
            raise ValueError("cpt_off does not match k[v] * prod k[parents]")
        if self.cpt.shape[0] != int(self.cpt_off[-1]):
            raise ValueError("cpt length != cpt_off[n]")

    def expected_cpt_sizes(self) -> np.ndarray:
        sizes = self.k.astype(np.int64).copy()
        if self.n_edges:
            kpar = self.k[self.in_idx].astype(np.int64)
            nz = np.diff(self.in_ptr) > 0
            # empty segments own no entries, so the non-empty starts delimit exactly
            sizes[nz] *= np.multiply.reduceat(kpar, self.in_ptr[:-1][nz])
        return sizes

    # ---- metric helpers -----------------------------------
    def messages_per_sweep(self) -> int:
        """One pi-message and one lambda-message per directed edge."""
        return 2 * self.n_edges

    def algorithmic_bytes_per_sweep(self) -> int:
        """fp64; CPT read once per node; every message / node vector read once and written once."""
        cpt_b = 8 * int(self.cpt_off[-1])
        vec = 2 * int(self.k[self.in_idx].sum()) if self.n_edges else 0
        vec += 2 * int(self.k.sum())
        return cpt_b + 16 * vec

    # ---- text dump for oracle/ref_driver -------------------------------------------
    def to_bnflat_text(self) -> str:
        out = ["BNFLAT1", str(self.n), " ".join(map(str, self.k.tolist()))]
        for v in range(self.n):
            p = self.parents(v)
            out.append(" ".join([str(p.size)] + [str(int(x)) for x in p]))
        for v in range(self.n):
            out.append(" ".join(repr(float(x)) for x in self.cpt[self.cpt_off[v]:self.cpt_off[v + 1]]))
        return "\n".join(out) + "\n"


def from_parent_lists(k, parents, cpts, name="") -> FlatModel:
    """Build a FlatModel from per-node python lists (small hand-written networks)."""
    n = len(k)
    in_ptr = np.zeros(n + 1, dtype=np.int32)
    in_idx = []
    cpt_off = np.zeros(n + 1, dtype=np.int64)
    flat = []
    for v in range(n):
        ps = list(parents[v])
        if ps != sorted(ps):
            raise ValueError("parents must be listed in ascending node id")
        in_idx.extend(ps)
        in_ptr[v + 1] = len(in_idx)
        c = np.asarray(cpts[v], dtype=np.float64).reshape(-1)
        flat.append(c)
        cpt_off[v + 1] = cpt_off[v] + c.size
    m = FlatModel(np.asarray(k, dtype=np.int32), in_ptr, np.asarray(in_idx, dtype=np.int32), cpt_off,
                  np.concatenate(flat) if flat else np.zeros(0), name=name)
    m.validate()
    return m


@dataclass
class Evidence:
    """BP evidence: a 1 x k[v] vector per node (one-hot in every reference test; soft allowed).

    Mirrors ``std::unordered_map<vertex_type, matrix_type> precondition``
    (``belief_propagation.hpp:31``)."""
    node: np.ndarray   # int32 [ne]
    off: np.ndarray    # int32 [ne+1]
    val: np.ndarray    # float64 [sum k]

    def __post_init__(self):
        self.node = np.ascontiguousarray(self.node, dtype=np.int32)
        self.off = np.ascontiguousarray(self.off, dtype=np.int32)
        self.val = np.ascontiguousarray(self.val, dtype=np.float64)

    @property
    def ne(self) -> int:
        return int(self.node.shape[0])

    @staticmethod
    def none() -> "Evidence":
        return Evidence(np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0))

    @staticmethod
    def from_dict(model: FlatModel, d) -> "Evidence":
        """d: {node: state_index | sequence of k[node] weights}."""
        nodes, off, vals = [], [0], []
        for v in sorted(d):
            x = d[v]
            kv = int(model.k[v])
            if np.isscalar(x):
                vec = np.zeros(kv)
                vec[int(x)] = 1.0
            else:
                vec = np.asarray(x, dtype=np.float64).reshape(-1)
                if vec.size != kv:
                    raise ValueError(f"evidence for node {v} needs {kv} entries")
            nodes.append(v)
            vals.append(vec)
            off.append(off[-1] + kv)
        return Evidence(np.asarray(nodes, np.int32), np.asarray(off, np.int32),
                        np.concatenate(vals) if vals else np.zeros(0))

    def hard_states(self, model: FlatModel) -> np.ndarray:
        """int32 [n], clamped state or -1 (likelihood-weighting style evidence)."""
        st = np.full(model.n, -1, dtype=np.int32)
        for j in range(self.ne):
            st[self.node[j]] = int(np.argmax(self.val[self.off[j]:self.off[j + 1]]))
        return st

    def to_bnflat_text(self) -> str:
        out = [str(self.ne)]
        for j in range(self.ne):
            v = self.val[self.off[j]:self.off[j + 1]]
            out.append(" ".join([str(int(self.node[j])), str(v.size)] + [repr(float(x)) for x in v]))
        return "\n".join(out) + "\n"

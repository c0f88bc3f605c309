"""Synthetic Bayesian networks for parity tests and bench.py.

Random numbers come from a stateless splitmix64 stream (Steele, Lea & Flood 2014):
the j-th 64-bit word of stream ``seed`` is ``mix(seed + (j+1) * 0x9E3779B97F4A7C15)``,
mapped to [0,1) as ``(x >> 11) * 2**-53``.  It is vectorisable in numpy and three lines of C,
so fixtures regenerate identically anywhere.  CPT rows are ``0.1 + 0.9 u`` divided by the row
sum: strictly positive, so the reference's unguarded normalise (``belief_propagation.hpp:298``)
never divides by zero.

Also holds the two hand-written networks of the reference's own BP tests
(``libs/bayesian/test/belief_propagation.cpp:9-62`` Pearl R,S->W,H and ``:127-182`` the
A->B->C->D "resume" chain) restated as flat models.
"""
from __future__ import annotations

import numpy as np

from .flat import Evidence, FlatModel, from_parent_lists

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, start: int, count: int) -> np.ndarray:
    """Words [start, start+count) of stream `seed` as uint64."""
    with np.errstate(over="ignore"):
        j = np.arange(start + 1, start + 1 + count, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + j * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, start: int, count: int) -> np.ndarray:
    return (splitmix64(seed, start, count) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _random_cpts(k: np.ndarray, in_ptr: np.ndarray, in_idx: np.ndarray, seed: int):
    """Strictly positive row-normalised CPTs for an arbitrary structure."""
    n = k.shape[0]
    rows = np.ones(n, dtype=np.int64)
    if in_idx.size:
        nz = np.diff(in_ptr) > 0
        rows[nz] = np.multiply.reduceat(k[in_idx].astype(np.int64), in_ptr[:-1][nz])
    sizes = rows * k
    cpt_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(sizes, out=cpt_off[1:])
    total = int(cpt_off[-1])
    if n and total >= (1 << 22) and int(k.min()) == int(k.max()):
        # uniform arity, large model (the 2048x2048 grid is 2.1 GB of CPTs): same numbers, built in
        # 128 MiB chunks with no whole-array temporaries.  np.add.reduceat over a run of kk entries
        # evaluates ((r1 + r2) + ... + r_{kk-1}) + r0, reproduced here term by term (asserted equal
        # to the general path in tests/test_host_logic.py).
        kk = int(k[0])
        cpt = np.empty(total, dtype=np.float64)
        chunk = (1 << 24) - (1 << 24) % kk
        for s0 in range(0, total, chunk):
            c = min(chunk, total - s0)
            r = uniform01(seed, s0, c)
            r *= 0.9
            r += 0.1
            q = r.reshape(-1, kk)
            sm = q[:, 1 if kk > 1 else 0].copy()
            for i in range(2, kk):
                sm += q[:, i]
            if kk > 1:
                sm += q[:, 0]
            q /= sm[:, None]
            cpt[s0:s0 + c] = r
        return cpt_off, cpt
    r = 0.1 + 0.9 * uniform01(seed, 0, total)
    # per-row sums: rows are contiguous runs of k[v] entries
    row_len = np.repeat(k.astype(np.int64), rows)
    row_start = np.zeros(row_len.shape[0], dtype=np.int64)
    np.cumsum(row_len[:-1], out=row_start[1:])
    s = np.add.reduceat(r, row_start)
    cpt = r / np.repeat(s, row_len)
    return cpt_off, cpt


def grid(rows: int, cols: int, k: int = 4, seed: int = 2, name: str | None = None) -> FlatModel:
    """2-D grid BN: node (r,c) <- (r-1,c), (r,c-1).  Config 3 is grid(316, 316, 4, seed=2)."""
    n = rows * cols
    ids = np.arange(n, dtype=np.int64)
    r, c = ids // cols, ids % cols
    deg = (r > 0).astype(np.int32) + (c > 0).astype(np.int32)
    in_ptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(deg, out=in_ptr[1:])
    in_idx = np.empty(int(in_ptr[-1]), dtype=np.int32)
    up, left = r > 0, c > 0
    # ascending parent id: (r-1,c) = id-cols comes before (r,c-1) = id-1 (cols >= 2)
    in_idx[in_ptr[:-1][up]] = (ids - cols)[up]
    in_idx[(in_ptr[:-1] + up.astype(np.int32))[left]] = (ids - 1)[left]
    if cols == 1:  # degenerate column: the single parent is id-1 = id-cols
        pass
    kk = np.full(n, k, dtype=np.int32)
    cpt_off, cpt = _random_cpts(kk, in_ptr, in_idx, seed)
    m = FlatModel(kk, in_ptr, in_idx, cpt_off, cpt, name=name or f"grid{rows}x{cols}_k{k}_s{seed}",
                  meta={"kind": "grid", "rows": rows, "cols": cols, "k": k, "seed": seed})
    return m


def random_dag(n: int, max_parents: int = 4, window: int = 64, k=4, seed: int = 1,
               name: str | None = None) -> FlatModel:
    """Random DAG: node i draws m_i uniform in {0..min(max_parents, i)} parents without
    replacement from [max(0, i-window), i).  `k` is an int or a sequence cycled over nodes
    (mixed arity).  Config 2 is random_dag(10000, 4, 64, 4, seed=1)."""
    draws = splitmix64(seed ^ 0x5DEECE66D, 0, n * (2 + 4 * max_parents) + 16)
    di = 0
    in_ptr = np.zeros(n + 1, dtype=np.int32)
    in_idx = []
    for i in range(n):
        lo = max(0, i - window)
        cap = min(max_parents, i - lo)
        m = int(draws[di] % np.uint64(cap + 1)); di += 1
        chosen = set()
        while len(chosen) < m:
            if di >= draws.shape[0]:
                draws = np.concatenate([draws, splitmix64(seed ^ 0x5DEECE66D, draws.shape[0], n)])
            chosen.add(lo + int(draws[di] % np.uint64(i - lo))); di += 1
        in_idx.extend(sorted(chosen))
        in_ptr[i + 1] = len(in_idx)
    in_idx = np.asarray(in_idx, dtype=np.int32)
    if np.isscalar(k):
        kk = np.full(n, int(k), dtype=np.int32)
    else:
        kk = np.resize(np.asarray(k, dtype=np.int32), n)
    cpt_off, cpt = _random_cpts(kk, in_ptr, in_idx, seed)
    return FlatModel(kk, in_ptr, in_idx, cpt_off, cpt, name=name or f"dag{n}_p{max_parents}_s{seed}",
                     meta={"kind": "dag", "n": n, "max_parents": max_parents, "window": window, "seed": seed})


def random_evidence(model: FlatModel, frac: float, seed: int = 7) -> Evidence:
    """floor(frac*V) distinct nodes by repeated x mod V, state x' mod k (one-hot vectors)."""
    want = int(np.floor(frac * model.n))
    if want == 0:
        return Evidence.none()
    words = splitmix64(seed ^ 0xE71DE9CE, 0, 4 * want + 64)
    chosen = {}
    i = 0
    while len(chosen) < want:
        if i + 1 >= words.shape[0]:
            words = np.concatenate([words, splitmix64(seed ^ 0xE71DE9CE, words.shape[0], 4 * want + 64)])
        v = int(words[i] % np.uint64(model.n))
        if v not in chosen:
            chosen[v] = int(words[i + 1] % np.uint64(int(model.k[v])))
        i += 2
    return Evidence.from_dict(model, chosen)


# ---- the reference's own test networks ---------------------------------------------

def pearl() -> FlatModel:
    """libs/bayesian/test/belief_propagation.cpp:9-62: R, S, W<-R, H<-R,S (all binary)."""
    return from_parent_lists(
        k=[2, 2, 2, 2],
        parents=[[], [], [0], [0, 1]],
        cpts=[[0.2, 0.8], [0.1, 0.9], [1.0, 0.0, 0.2, 0.8],
              [1.0, 0.0, 1.0, 0.0, 0.9, 0.1, 0.0, 1.0]],
        name="pearl")


def resume_chain() -> FlatModel:
    """libs/bayesian/test/belief_propagation.cpp:127-182: A->B->C->D, arities 3,3,2,3."""
    return from_parent_lists(
        k=[3, 3, 2, 3],
        parents=[[], [0], [1], [2]],
        cpts=[[0.30, 0.60, 0.10],
              [0.20, 0.30, 0.50, 0.30, 0.30, 0.40, 0.80, 0.10, 0.10],
              [0.50, 0.50, 0.70, 0.30, 0.40, 0.60],
              [0.40, 0.30, 0.30, 0.20, 0.60, 0.20]],
        name="resume_chain")

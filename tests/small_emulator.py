"""CPU emulation of bn_small.hip on the plan bn_small_plan.cpp builds (Engine.small_plan()): the same work items,
executed one after another in Python floats (IEEE doubles, same operation order).  Test infrastructure: it checks
the PLANNER -- indices, places, summation order -- against the oracle without a GPU."""
import numpy as np


def emulate(plan, model, evidence, eps, max_sweeps=0):
    """plan: Engine.small_plan() (one workgroup) or Engine.mid_plan() (a list: one plan per workgroup of bn_mid.hip -- message and
    node-vector indices are global there, so the parts simply run one after another inside an iteration: they read the old
    state only and write disjoint elements of the new one)."""
    parts = plan if isinstance(plan, list) else [plan]
    plan = parts[0]
    N, M = plan["N"], plan["M"]
    pi = np.ones((2, max(M, 1)))
    lam = np.ones((2, max(M, 1)))
    npi = np.zeros((2, N))
    nlam = np.zeros((2, N))
    node_off = np.concatenate([[0], np.cumsum(model.k)]).astype(np.int64)
    frz = np.zeros(N, dtype=bool)
    npi[0] = plan["npi_init"]
    nlam[0] = 1.0
    for j in range(evidence.ne):
        v = int(evidence.node[j])
        lo, hi = node_off[v], node_off[v + 1]
        frz[lo:hi] = True
        npi[0, lo:hi] = nlam[0, lo:hi] = evidence.val[evidence.off[j]:evidence.off[j + 1]]
    residuals = []
    s = 0

    def normalise_rows(slots, vals, buf_of):
        """slots [rounds*nt, 4]; vals per slot (un-normalised); returns normalised values per slot"""
        out = {}
        for q, val in vals.items():
            y = int(slots[q, 1])
            k, first = (y >> 16) & 0xff, y >> 24
            lane = q % 64
            vec0 = q - (lane - first)  # slot of the vector's element 0
            total = 0.0
            for r in range(k):
                total += vals[vec0 + r]
            with np.errstate(divide="ignore", invalid="ignore"):
                out[q] = np.float64(val) / np.float64(total)
        return out

    while True:
        cur = s & 1
        new = cur ^ 1
        md = 0.0
        for plan in parts:
            ent, cpt, term, clist = plan["ent"], plan["ent_cpt"], plan["term"], plan["clist"]
            stg = np.zeros(plan["T"])
            # phase 1: entry items
            for e in range(ent.shape[0]):
                x, y = int(ent[e, 0]), int(ent[e, 1])
                if not (y >> 24) & 1:
                    continue
                m, tbase = (y >> 16) & 0xff, y & 0xffff
                li = nlam[cur, x & 0xffff]
                tw = [int(term[tbase + j]) for j in range(m)]
                pj = [pi[cur, t & 0xffff] for t in tw]
                c = np.float64(cpt[e])
                v = c
                for j in range(m):
                    v = v * pj[j]
                stg[x >> 16] = v
                lc = li * c
                for jt in range(m):
                    w = lc
                    for j in range(m):
                        if j != jt:
                            w = w * pj[j]
                    stg[tw[jt] >> 16] = w
            # phase 2a: accumulator items
            bvals = {}
            bs = plan["bslot"]
            for q in range(bs.shape[0]):
                if bs[q, 2] == 0:
                    continue
                base, n8 = int(bs[q, 0]) & 0xffff, int(bs[q, 0]) >> 16
                acc = 0.0
                for r in range(n8):
                    acc = acc + stg[base + r]
                bvals[q] = acc
            bn = normalise_rows(bs, bvals, None)
            for q, val in bn.items():
                kind, out_idx = int(bs[q, 2]) & 0xff, int(bs[q, 1]) & 0xffff
                if kind == 1:
                    npi[new, out_idx] = npi[cur, out_idx] if frz[out_idx] else val
                else:
                    lam[new, out_idx] = val
                    d = abs(val - lam[cur, out_idx])
                    md = d if md < d else md
            # phase 2b: product items
            cvals = {}
            cs = plan["cslot"]
            for q in range(cs.shape[0]):
                kind = int(cs[q, 2]) & 0xff
                if kind == 0:
                    continue
                skip = (int(cs[q, 2]) >> 8) & 0xffff
                cl, deg = int(cs[q, 0]) & 0xffff, int(cs[q, 0]) >> 16
                at = q % 64 - (int(cs[q, 1]) >> 24)
                val = npi[cur, int(cs[q, 3]) & 0xffff] if kind == 4 else 1.0
                for xq in range(deg):
                    if xq != skip:
                        val = val * lam[cur, int(clist[cl + xq]) + at]
                cvals[q] = val
            cn = normalise_rows(cs, cvals, None)
            for q, val in cn.items():
                kind, out_idx = int(cs[q, 2]) & 0xff, int(cs[q, 1]) & 0xffff
                if kind == 3:
                    nlam[new, out_idx] = nlam[cur, out_idx] if frz[out_idx] else val
                else:
                    pi[new, out_idx] = val
                    d = abs(val - pi[cur, out_idx])
                    md = d if md < d else md
        md = max(md, np.finfo(np.float64).tiny)
        residuals.append(md)
        s += 1
        if md < eps or (max_sweeps > 0 and s >= max_sweeps):
            break
    fin = s & 1
    beliefs = np.zeros(N)
    for v in range(model.n):
        lo, hi = node_off[v], node_off[v + 1]
        b = [npi[fin, i] * nlam[fin, i] for i in range(lo, hi)]
        total = 0.0
        for x in b:
            total = total + x
        with np.errstate(divide="ignore", invalid="ignore"):
            beliefs[lo:hi] = np.array(b) / np.float64(total)
    return {"beliefs": beliefs, "sweeps": s, "residuals": np.array(residuals), "pi_msg": pi[fin, :M].copy(), "lambda_msg": lam[fin, :M].copy()}

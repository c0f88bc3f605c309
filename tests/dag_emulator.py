"""CPU emulation of bn_dag.hip on the plan bn_dag_plan.cpp builds (Engine.dag_plan()): the same tiles, the same lanes, the
same operation order -- the factored contraction of the lane-group tiles included, with its shuffle butterflies -- in numpy
doubles, 64 lanes at a time.  Test infrastructure: it checks the PLANNER (CPT image, lane digits, edge ids, item order)
and the arithmetic form against the oracle without a GPU."""
import numpy as np

K = 4
LANES = np.arange(64)


def _norm(t):
    """normalize (:298-311): plain left-to-right sum, no zero guard; t [..., 4]"""
    s = ((t[..., 0] + t[..., 1]) + t[..., 2]) + t[..., 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        return t / s[..., None]


def _resmax(md, d):
    """std::max(md, d): a NaN d is dropped"""
    d = d[~np.isnan(d)]
    return max(md, float(d.max())) if d.size else md


class _State:
    def __init__(self, n, E):
        self.pim = np.ones((2, max(E, 1), K))
        self.lam = np.ones((2, max(E, 1), K))
        self.npi = np.zeros((2, n, K))
        self.nlam = np.zeros((2, n, K))
        self.frz = np.zeros(n, dtype=bool)


def _cpt_of(plan, tile, entries):
    """[64 lanes, entries] from the tile's image: entry pair q of lane l at double2 cpt_base + q * 64 + l"""
    base = int(tile[3])
    img = plan["cpt_img"]
    q = np.arange(entries)
    idx = (base + (q[None, :] >> 1) * 64 + LANES[:, None]) * 2 + (q[None, :] & 1)
    return img[idx]


def _child_u(plan, st, tile, M, s):
    first = s == 0 and not plan.get("state_init", False)   # (a padded network's initial state stands in memory)
    cur, nxt = s & 1, (s & 1) ^ 1
    C = K ** M
    cn = plan["cnode"][int(tile[2]):int(tile[2]) + 64]
    active = cn[:, 0] >= 0
    node = np.where(active, cn[:, 0], 0)
    eb = np.where(active, cn[:, 1], 0)
    frozen = active & st.frz[node]
    cpt = _cpt_of(plan, tile, K * C)  # entry cond * 4 + i
    pim = np.ones((64, max(M, 1), K))
    lold = np.ones((64, max(M, 1), K))
    if not first:
        for j in range(M):
            pim[:, j] = st.pim[cur, eb + j]
            lold[:, j] = st.lam[cur, eb + j]
    lav = np.ones((64, K))
    pold = np.ones((64, K))
    ld = np.full(64, not first) | frozen
    lav[ld] = st.nlam[cur, node[ld]]
    pold[ld] = st.npi[cur, node[ld]]
    pin = np.zeros((64, K))
    out = np.zeros((64, max(M, 1), K))
    for ib in range(K):
        if M == 0:
            pin[:, ib] = 0.0 + cpt[:, ib]
            continue
        acc = np.zeros(64)
        for rr in range(C // K):
            for x in range(K):
                cond = rr * K + x
                v = cpt[:, cond * K + ib].copy()
                for j in range(M):
                    v = v * pim[:, j, (cond // K ** (M - 1 - j)) % K]
                acc = acc + v
            for jt in range(M):
                stride = K ** (M - 1 - jt)
                for ct in range(K):
                    cond = (rr // stride) * stride * K + ct * stride + (rr % stride)
                    v = lav[:, ib] * cpt[:, cond * K + ib]
                    for j in range(M):
                        if j != jt:
                            v = v * pim[:, j, (cond // K ** (M - 1 - j)) % K]
                    out[:, jt, ct] = out[:, jt, ct] + v
        pin[:, ib] = acc
    pin = _norm(pin)
    md = 0.0
    for jt in range(M):
        o = _norm(out[:, jt])
        md = _resmax(md, np.abs(o - lold[:, jt])[active].ravel())
        st.lam[nxt, eb[active] + jt] = o[active]
    st.npi[nxt, node[active]] = np.where(frozen[active, None], pold[active], pin[active])
    return md


def _child_g(plan, st, tile, D, s):
    first = s == 0 and not plan.get("state_init", False)   # (a padded network's initial state stands in memory)
    cur, nxt = s & 1, (s & 1) ^ 1
    M, G = D + 2, 4 ** D
    cn = plan["cnode"][int(tile[2]):int(tile[2]) + 64]
    active = cn[:, 0] >= 0
    node = np.where(active, cn[:, 0], 0)
    eb = np.where(active, cn[:, 1], 0)
    frozen = active & st.frz[node]
    g, nl = LANES % G, LANES // G
    cpt = _cpt_of(plan, tile, 64).reshape(64, K, K, K)  # [lane, c, d, i]
    pim = np.ones((64, M, K))
    if not first:
        for j in range(M):
            pim[:, j] = st.pim[cur, eb + j]
    lav = np.ones((64, K))
    ld = np.full(64, not first) | frozen
    lav[ld] = st.nlam[cur, node[ld]]
    fold = np.ones((64, K))
    fin_msg = (g >= 1) & (g <= M)
    if not first:
        fold[fin_msg] = st.lam[cur, eb[fin_msg] + g[fin_msg] - 1]
    sel = (g == 0) & frozen
    fold[sel] = st.npi[cur, node[sel]]
    pfix = [pim[LANES, j, (g >> (2 * (D - 1 - j))) & 3] for j in range(D)]
    pC, pD = pim[:, D], pim[:, D + 1]
    S = np.zeros((64, K)); LC = np.zeros((64, K)); LD = np.zeros((64, K))
    for c in range(K):
        R = np.zeros((64, K)); lc = np.zeros(64)
        for d in range(K):
            e = cpt[:, c, d]
            q = (lav[:, 0] * e[:, 0] + lav[:, 1] * e[:, 1]) + (lav[:, 2] * e[:, 2] + lav[:, 3] * e[:, 3])
            for i in range(K):
                R[:, i] = R[:, i] + e[:, i] * pD[:, d]
            lc = lc + q * pD[:, d]
            LD[:, d] = LD[:, d] + pC[:, c] * q
        for i in range(K):
            S[:, i] = S[:, i] + pC[:, c] * R[:, i]
        LC[:, c] = lc
    L = np.zeros(64)
    for c in range(K):
        L = L + pC[:, c] * LC[:, c]
    w = np.ones(64)
    for j in range(D):
        w = w * pfix[j]
    pp = w[:, None] * S
    ol = [w[:, None] * LC, w[:, None] * LD]
    sf = []
    for jt in range(D):
        x = L.copy()
        for j in range(D):
            if j != jt:
                x = x * pfix[j]
        sf.append(x)
    mask = 1
    while mask < G:
        pp = pp + pp[LANES ^ mask]
        ol = [o + o[LANES ^ mask] for o in ol]
        mask <<= 1
    of = []
    for jt in range(D):
        x = sf[jt]
        for j in range(D):
            if j != jt:
                x = x + x[LANES ^ (1 << (2 * (D - 1 - j)))]
                x = x + x[LANES ^ (2 << (2 * (D - 1 - j)))]
        of.append(np.stack([x[nl * G + (ct << (2 * (D - 1 - jt)))] for ct in range(K)], axis=1))
    o = pp.copy()
    for jt in range(M):
        pick = g == jt + 1
        src = of[jt] if jt < D else ol[jt - D]
        o[pick] = src[pick]
    o = _norm(o)
    md = 0.0
    fin = active & (g <= M)
    m0 = fin & (g == 0)
    st.npi[nxt, node[m0]] = np.where(frozen[m0, None], fold[m0], o[m0])
    mm = fin & (g >= 1)
    md = _resmax(md, np.abs(o - fold)[mm].ravel())
    st.lam[nxt, eb[mm] + g[mm] - 1] = o[mm]
    return md


def _parent(plan, st, tile, s):
    first = s == 0 and not plan.get("state_init", False)   # (a padded network's initial state stands in memory)
    cur, nxt = s & 1, (s & 1) ^ 1
    it = plan["pitem"][int(tile[2]):int(tile[2]) + 64]
    active = it[:, 0] >= 0
    node = np.where(active, it[:, 0], 0)
    tedge, obeg = it[:, 1], it[:, 2]
    deg = np.where(active, it[:, 3] & 0xffff, 0)
    tpos = (it[:, 3] >> 16) & 0xffff
    dmax = int(tile[4])
    frozen = active & st.frz[node]
    is_msg = tedge >= 0
    acc = (np.arange(K)[None, :] < plan["node_k"][node][:, None]).astype(np.float64)
    old = np.ones((64, K))
    a = is_msg & (np.full(64, not first) | frozen)
    acc[a] = st.npi[cur, node[a]]
    b = is_msg & ~a
    acc[b] = plan["npi_init"][node[b]]
    if not first:
        old[is_msg] = st.pim[cur, tedge[is_msg]]
    c = ~is_msg & frozen
    old[c] = st.nlam[cur, node[c]]
    oedge = plan["oedge"]
    for x in range(dmax):
        lk = np.ones((64, K))
        has = x < deg
        if not first and has.any():
            lk[has] = st.lam[cur, oedge[obeg[has] + x]]
        use = (x != tpos)
        acc = acc * np.where(use[:, None], lk, 1.0)
    acc = _norm(acc)
    md = 0.0
    m = active & is_msg
    md = _resmax(md, np.abs(acc - old)[m].ravel())
    st.pim[nxt, tedge[m]] = acc[m]
    nm = active & ~is_msg
    st.nlam[nxt, node[nm]] = np.where(frozen[nm, None], old[nm], acc[nm])
    return md


def _parent_packed(plan, st, tile, s):
    """kDagParent: a node's c + 1 items in adjacent lanes; every lane loads ONE record, the node's lanes read each other's"""
    first = s == 0 and not plan.get("state_init", False)   # (a padded network's initial state stands in memory)
    cur, nxt = s & 1, (s & 1) ^ 1
    it = plan["pitem"][int(tile[2]):int(tile[2]) + 64]
    active = it[:, 0] >= 0
    node = np.where(active, it[:, 0], 0)
    tedge = np.where(active, it[:, 1], -1)
    deg = np.where(active, it[:, 3] & 0xffff, 0)
    tpos = np.where(active, it[:, 3] >> 16, -1)
    first_lane = LANES - (tpos + 1)
    dmax = int(tile[4])
    frozen = active & st.frz[node]
    is_msg = tedge >= 0
    rec = np.ones((64, K))
    old = np.ones((64, K))
    if not first:
        rec[is_msg] = st.lam[cur, tedge[is_msg]]
        old[is_msg] = st.pim[cur, tedge[is_msg]]
    a = ~is_msg & (np.full(64, not first) | frozen)
    rec[a] = st.npi[cur, node[a]]
    b = ~is_msg & ~a & active
    rec[b] = plan["npi_init"][node[b]]
    c = ~is_msg & frozen
    old[c] = st.nlam[cur, node[c]]
    assert (first_lane >= 0).all() and (first_lane[active] + deg[active] <= 63).all()
    acc = np.where(is_msg[:, None], rec[first_lane], (np.arange(K)[None, :] < plan["node_k"][node][:, None]).astype(np.float64))
    for x in range(dmax):
        src = np.where(x < deg, first_lane + 1 + x, LANES)
        use = (x < deg) & (x != tpos)
        acc = acc * np.where(use[:, None], rec[src], 1.0)
    acc = _norm(acc)
    md = 0.0
    m = active & is_msg
    md = _resmax(md, np.abs(acc - old)[m].ravel())
    st.pim[nxt, tedge[m]] = acc[m]
    nm = active & ~is_msg
    st.nlam[nxt, node[nm]] = np.where(frozen[nm, None], old[nm], acc[nm])
    return md


def emulate(plan, model, evidence, eps, max_sweeps=0):
    n, E = plan["n"], plan["E"]
    st = _State(n, E)
    k = np.asarray(model.k, dtype=np.int64)
    plan = dict(plan, node_k=k, state_init=bool((k != K).any()))
    for j in range(evidence.ne):
        v = int(evidence.node[j])
        vec = np.zeros(K)
        vec[:k[v]] = evidence.val[evidence.off[j]:evidence.off[j + 1]]
        st.npi[:, v] = vec
        st.nlam[:, v] = vec
        st.frz[v] = True
    if plan["state_init"]:   # dag_init_kernel: ones over the states that exist, zeros in the padding; evidence nodes keep their vectors
        kp = k[np.asarray(model.in_idx, dtype=np.int64)] if E else np.zeros(0, np.int64)
        st.pim[0, :E] = st.lam[0, :E] = (np.arange(K)[None, :] < kp[:, None]).astype(np.float64)
        free = ~st.frz
        st.npi[0, free] = plan["npi_init"][free]
        st.nlam[0, free] = (np.arange(K)[None, :] < k[free][:, None]).astype(np.float64)
    residuals = []
    s = 0
    tiny = np.finfo(np.float64).tiny
    while True:
        md = 0.0
        for t in plan["tiles"]:
            kind = int(t[0])
            if kind <= 2:
                md = max(md, _child_u(plan, st, t, kind, s))
            elif kind <= 5:
                md = max(md, _child_g(plan, st, t, kind - 2, s))
            elif kind == 8:
                md = max(md, _parent_packed(plan, st, t, s))
            else:
                md = max(md, _parent(plan, st, t, s))
        md = max(md, tiny)
        residuals.append(md)
        s += 1
        if md < eps or (max_sweeps > 0 and s >= max_sweeps):
            break
    fin = s & 1
    bel = st.npi[fin] * st.nlam[fin]
    exists = np.arange(K)[None, :] < k[:, None]
    beliefs = _norm(bel)[exists]
    kp = k[np.asarray(model.in_idx, dtype=np.int64)] if E else np.zeros(0, np.int64)
    ex_e = np.arange(K)[None, :] < kp[:, None]
    return {"beliefs": beliefs, "sweeps": s, "residuals": np.array(residuals),
            "pi_msg": st.pim[fin, :E][ex_e].copy(), "lambda_msg": st.lam[fin, :E][ex_e].copy()}

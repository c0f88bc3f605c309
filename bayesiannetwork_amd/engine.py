"""Python host mirror of the reference's inference functors over the C ABI.

``BeliefPropagation(model)(evidence, epsilon)`` mirrors
``bn::inference::belief_propagation`` (reference ``bayesian/inference/belief_propagation.hpp:12-31``:
constructed from the network, called with evidence and epsilon = 0.001, returns per-node 1 x k
marginals); ``LikelihoodWeighting(model)(evidence, sample_num)`` mirrors
``bn::inference::likelihood_weighting`` (``likelihood_weighting.hpp:13-59``, sample_num = 10000).
All compute happens in the HIP library; numpy only marshals arrays.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .flat import Evidence, FlatModel


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


class Engine:
    """One flattened network resident on one GPU (bn_create / bn_destroy)."""

    def __init__(self, model: FlatModel, device: int = _lib.BN_DEVICE_CURRENT, lanes_per_node: int = 0,
                 rank: int = 0, nranks: int = 1, owner=None):
        """rank/nranks/owner: shard `rank` of an edge-cut partition (bn_create_sharded);
        owner[v] in [0, nranks), None = balanced contiguous node ranges."""
        self.model = model
        self._nbel = int(model.k.sum())   # doubles of a node-major result (summing k per call cost 28 us on the 99 856-node grid: 12 % of a query)
        self.rank, self.nranks = rank, nranks
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        d = _lib.ModelDesc(model.n, _p(model.k, ctypes.c_int32), _p(model.in_ptr, ctypes.c_int32),
                           _p(model.in_idx, ctypes.c_int32), _p(model.cpt_off, ctypes.c_int64),
                           _p(model.cpt, ctypes.c_double), device, lanes_per_node)
        if nranks == 1:
            _lib.check(L.bn_create(ctypes.byref(d), ctypes.byref(self._h)))
        else:
            own = None if owner is None else np.ascontiguousarray(owner, dtype=np.int32)
            _lib.check(L.bn_create_sharded(ctypes.byref(d), rank, nranks,
                                           None if own is None else _p(own, ctypes.c_int32), ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().bn_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- belief propagation ------------------------------------------------------------
    def bp_set_evidence(self, evidence: Evidence | None = None) -> None:
        """Validate + upload an evidence set once (bn_bp_set_evidence); it stays resident in HBM."""
        ev = evidence if evidence is not None else Evidence.none()
        _lib.check(_lib.lib().bn_bp_set_evidence(self._h, ev.ne, _p(ev.node, ctypes.c_int32),
                                                 _p(ev.off, ctypes.c_int32), _p(ev.val, ctypes.c_double)))

    def bp_run_device(self, eps: float = 0.001, max_sweeps: int = 0):
        """Run on the staged evidence; beliefs stay in device memory (bn_bp_run_device)."""
        sweeps = ctypes.c_int32(0)
        res = ctypes.c_double(0.0)
        _lib.check(_lib.lib().bn_bp_run_device(self._h, float(eps), int(max_sweeps), ctypes.byref(sweeps),
                                               ctypes.byref(res)))
        return {"beliefs": None, "sweeps": sweeps.value, "residual": res.value}

    def bp_run(self, evidence: Evidence | None = None, eps: float = 0.001, max_sweeps: int = 0):
        """Evidence in, host beliefs out (bn_bp_run)."""
        ev = evidence if evidence is not None else Evidence.none()
        sweeps = ctypes.c_int32(0)
        res = ctypes.c_double(0.0)
        bel = np.empty(self._nbel, dtype=np.float64)
        _lib.check(_lib.lib().bn_bp_run(self._h, ev.ne, _p(ev.node, ctypes.c_int32), _p(ev.off, ctypes.c_int32),
                                        _p(ev.val, ctypes.c_double), float(eps), int(max_sweeps),
                                        _p(bel, ctypes.c_double), ctypes.byref(sweeps), ctypes.byref(res)))
        return {"beliefs": bel, "sweeps": sweeps.value, "residual": res.value}

    def bp_run_view(self, evidence: Evidence | None = None, eps: float = 0.001, max_sweeps: int = 0):
        """bn_bp_run_view: like bp_run, but "beliefs" is a numpy VIEW of the engine's page-locked host buffer
        (no second copy); it is overwritten by the next run on this engine and dies with it -- copy to keep."""
        ev = evidence if evidence is not None else Evidence.none()
        sweeps = ctypes.c_int32(0)
        res = ctypes.c_double(0.0)
        view = _lib.f64p()
        # (the three pointers are made once per Evidence object and the array over the engine's buffer once per buffer address:
        # `ndarray.ctypes.data_as` and `np.ctypeslib.as_array` together cost ~8 us of a 200 us query)
        ptrs = getattr(ev, "_ptrs", None)
        if ptrs is None or ptrs[3] is not ev.node or ptrs[4] is not ev.off or ptrs[5] is not ev.val:   # (arrays replaced since)
            ptrs = ev._ptrs = (_p(ev.node, ctypes.c_int32), _p(ev.off, ctypes.c_int32), _p(ev.val, ctypes.c_double), ev.node, ev.off, ev.val)
        _lib.check(_lib.lib().bn_bp_run_view(self._h, ev.ne, ptrs[0], ptrs[1], ptrs[2], float(eps), int(max_sweeps),
                                             ctypes.byref(view), ctypes.byref(sweeps), ctypes.byref(res)))
        n = self._nbel
        if not n:
            return {"beliefs": np.zeros(0), "sweeps": sweeps.value, "residual": res.value}
        addr = ctypes.cast(view, ctypes.c_void_p).value
        cached = getattr(self, "_view_cache", None)
        if cached is None or cached[0] != addr:
            cached = self._view_cache = (addr, np.ctypeslib.as_array(view, shape=(n,)))
        return {"beliefs": cached[1], "sweeps": sweeps.value, "residual": res.value}

    # ---- several evidence sets per call (bn_bp_*_batch) ------------------------------------
    @staticmethod
    def _pack_sets(evidences):
        evs = [ev if ev is not None else Evidence.none() for ev in evidences]
        ne = np.asarray([ev.ne for ev in evs], dtype=np.int32)
        node = np.concatenate([np.asarray(ev.node, np.int32) for ev in evs] + [np.zeros(0, np.int32)]).astype(np.int32)
        off = np.concatenate([np.asarray(ev.off, np.int32)[:ev.ne + 1] if ev.ne else np.zeros(1, np.int32) for ev in evs]).astype(np.int32)
        val = np.concatenate([np.asarray(ev.val, np.float64) for ev in evs] + [np.zeros(0)]).astype(np.float64)
        return ne, np.ascontiguousarray(node), np.ascontiguousarray(off), np.ascontiguousarray(val)

    def bp_set_evidence_batch(self, evidences) -> None:
        ne, node, off, val = self._pack_sets(evidences)
        self._n_sets = int(ne.size)
        _lib.check(_lib.lib().bn_bp_set_evidence_batch(self._h, ne.size, _p(ne, ctypes.c_int32), _p(node, ctypes.c_int32),
                                                       _p(off, ctypes.c_int32), _p(val, ctypes.c_double)))

    def bp_run_batch_device(self, eps: float = 0.001, max_sweeps: int = 0):
        sweeps = np.zeros(self._n_sets, dtype=np.int32)
        res = np.zeros(self._n_sets, dtype=np.float64)
        _lib.check(_lib.lib().bn_bp_run_batch_device(self._h, float(eps), int(max_sweeps), _p(sweeps, ctypes.c_int32),
                                                     _p(res, ctypes.c_double)))
        return {"sweeps": sweeps, "residual": res}

    def bp_beliefs_batch(self) -> np.ndarray:
        bel = np.empty((self._n_sets, self._nbel), dtype=np.float64)
        _lib.check(_lib.lib().bn_bp_copy_beliefs_batch(self._h, _p(bel, ctypes.c_double)))
        return bel

    def bp_residuals_batch(self, set_index: int, cap: int = 65536) -> np.ndarray:
        out = np.zeros(cap, dtype=np.float64)
        cnt = _lib.check(_lib.lib().bn_bp_residual_history_batch(self._h, int(set_index), _p(out, ctypes.c_double), cap))
        return out[:cnt].copy()

    def bp_run_batch(self, evidences, eps: float = 0.001, max_sweeps: int = 0):
        """Several evidence sets on this network in one call; every set gets the result of running it alone."""
        self.bp_set_evidence_batch(evidences)
        out = self.bp_run_batch_device(eps, max_sweeps)
        out["beliefs"] = self.bp_beliefs_batch()
        return out

    def bp_beliefs(self) -> np.ndarray:
        bel = np.empty(self._nbel, dtype=np.float64)
        _lib.check(_lib.lib().bn_bp_copy_beliefs(self._h, _p(bel, ctypes.c_double)))
        return bel

    def bp_residuals(self, cap: int = 65536) -> np.ndarray:
        out = np.zeros(cap, dtype=np.float64)
        cnt = _lib.check(_lib.lib().bn_bp_residual_history(self._h, _p(out, ctypes.c_double), cap))
        return out[:cnt].copy()

    def bp_messages(self):
        nm = int(self.model.k[self.model.in_idx].sum()) if self.model.n_edges else 0
        pi, lam = np.zeros(max(nm, 1)), np.zeros(max(nm, 1))
        _lib.check(_lib.lib().bn_bp_messages(self._h, _p(pi, ctypes.c_double), _p(lam, ctypes.c_double)))
        return pi[:nm], lam[:nm]

    def bp_stats(self) -> dict:
        st = _lib.BpStats()
        _lib.check(_lib.lib().bn_bp_last_stats(self._h, ctypes.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    def reload_cpt(self, cpt) -> None:
        """bn_reload_cpt: new CPT values (the whole flat array, the model's layout) on the unchanged structure."""
        cpt = np.ascontiguousarray(cpt, dtype=np.float64)
        _lib.check(_lib.lib().bn_reload_cpt(self._h, _p(cpt, ctypes.c_double), cpt.shape[0]))
        self.model = FlatModel(self.model.k, self.model.in_ptr, self.model.in_idx, self.model.cpt_off, cpt.copy(), name=self.model.name)

    def set_option(self, name: str, value: int) -> None:
        _lib.check(_lib.lib().bn_set_option(self._h, name.encode(), int(value)))

    def info(self, name: str) -> int:
        """bn_get_info: "flow_eligible", "last_flow", "nbr_max", "resident_eligible", "resident_blocks", "resident_aborts",
        "small_eligible", "small_waves", "small_lds_bytes" ..."""
        return _lib.check(_lib.lib().bn_get_info(self._h, name.encode()))

    def last_path(self) -> int:
        """0: one launch per sweep; 2: resident tiles; 3: one workgroup (small networks); 4: several workgroups (mid-size);
        5: register-resident child tiles + parent items (k = 4 networks with <= 5 parents)."""
        return _lib.check(_lib.lib().bn_bp_last_path(self._h))

    # ---- multi-GPU ---------------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.lib().bn_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes) -> None:
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        _lib.check(_lib.lib().bn_comm_init(self._h, buf))

    def peer_export(self) -> bytes:
        """bn_peer_export: this shard's blob for the in-kernel halo exchange (ship it to every rank)."""
        n = _lib.check(_lib.lib().bn_peer_blob_size(self._h))
        buf = ctypes.create_string_buffer(int(n))
        _lib.check(_lib.lib().bn_peer_export(self._h, buf, n))
        return buf.raw

    def peer_import(self, blobs) -> bool:
        """bn_peer_import: blobs[r] = rank r's export (own included).  True when the in-kernel exchange is set up."""
        keep = [ctypes.create_string_buffer(bytes(b), len(b)) for b in blobs]
        ptrs = (ctypes.c_void_p * len(keep))(*[ctypes.cast(k, ctypes.c_void_p) for k in keep])
        sizes = np.asarray([len(b) for b in blobs], dtype=np.int64)
        _lib.check(_lib.lib().bn_peer_import(self._h, ptrs, _p(sizes, ctypes.c_int64), len(keep)))
        return bool(self.info("shard_flow"))

    def flow_tables(self):
        """(neighbour slots [n_tiles, chunks * 64], report masks [n_tiles]) of the dataflow form (bn_layout_flow)."""
        nt, ch = self.layout()["n_tiles"], self.info("nbr_chunks")
        nbr = np.full(max(nt * ch * 64, 1), -1, dtype=np.int32)
        pub = np.zeros(max(nt, 1), dtype=np.uint32)
        _lib.check(_lib.lib().bn_layout_flow(self._h, _p(nbr, ctypes.c_int32), pub.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
        return nbr[:nt * ch * 64].reshape(nt, ch * 64) if ch else np.zeros((nt, 0), np.int32), pub[:nt]

    def mid_plan(self):
        """The plan that spreads a mid-size network over several workgroups (bn_mid_plan_get): a list with one dict per part, laid
        out like small_plan()'s plus v0, v1 (the part's node range); None when the network is not eligible."""
        if not self.info("mid_eligible"):
            return None
        L = _lib.lib()
        u32 = ctypes.POINTER(ctypes.c_uint32)
        parts = []
        for q in range(self.info("mid_parts")):
            dims = np.zeros(14, dtype=np.int32)
            _lib.check(L.bn_mid_plan_get(self._h, q, _p(dims, ctypes.c_int32), None, None, None, None, None, None, None))
            names = ["n", "N", "M", "S", "T", "TT", "CL", "waves", "re", "rb", "rc", "mmax", "v0", "v1"]
            d = {k: int(v) for k, v in zip(names, dims)}
            nt = 64 * d["waves"]
            ent = np.zeros((d["re"] * nt, 2), dtype=np.uint32)
            ent_cpt = np.zeros(d["re"] * nt, dtype=np.float64)
            term = np.zeros(max(d["TT"], 1), dtype=np.uint32)
            clist = np.zeros(max(d["CL"], 1), dtype=np.uint16)
            bslot = np.zeros((d["rb"] * nt, 4), dtype=np.uint32)
            cslot = np.zeros((d["rc"] * nt, 4), dtype=np.uint32)
            init = np.zeros(d["N"], dtype=np.float64)
            _lib.check(L.bn_mid_plan_get(self._h, q, None, ent.ctypes.data_as(u32), _p(ent_cpt, ctypes.c_double), term.ctypes.data_as(u32),
                                         clist.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), bslot.ctypes.data_as(u32),
                                         cslot.ctypes.data_as(u32), _p(init, ctypes.c_double)))
            d.update(ent=ent, ent_cpt=ent_cpt, term=term[:d["TT"]], clist=clist[:d["CL"]], bslot=bslot, cslot=cslot, npi_init=init)
            parts.append(d)
        return parts

    def dag_plan(self):
        """The plan of the register-resident DAG path (bn_dag_plan_get; k = 4 networks with <= 5 parents per node), or None when
        the network is not eligible: dict(n, E, n_tiles, blocks, stream, child_tiles, parent_tiles, tiles [n_tiles, 8], slot_ptr,
        cnode [n_tiles * 64, 2], pitem [n_tiles * 64, 4], oedge, cpt_img, npi_init [n, 4])."""
        if not self.info("dag_eligible"):
            return None
        L = _lib.lib()
        dims = np.zeros(8, dtype=np.int32)
        _lib.check(L.bn_dag_plan_get(self._h, _p(dims, ctypes.c_int32), None, None, None, None, None, None, None))
        names = ["n", "E", "n_tiles", "blocks", "stream", "child_tiles", "parent_tiles", "cpt_doubles"]
        d = {k: int(v) for k, v in zip(names, dims)}
        tiles = np.zeros((d["n_tiles"], 8), dtype=np.int32)
        slot_ptr = np.zeros(d["blocks"] * 8 + 1, dtype=np.int32)
        cnode = np.zeros((d["n_tiles"] * 64, 2), dtype=np.int32)
        pitem = np.zeros((d["n_tiles"] * 64, 4), dtype=np.int32)
        oedge = np.zeros(max(d["E"], 1), dtype=np.int32)
        img = np.zeros(max(d["cpt_doubles"], 1), dtype=np.float64)
        init = np.zeros((d["n"], 4), dtype=np.float64)
        _lib.check(L.bn_dag_plan_get(self._h, None, _p(tiles, ctypes.c_int32), _p(slot_ptr, ctypes.c_int32), _p(cnode, ctypes.c_int32),
                                     _p(pitem, ctypes.c_int32), _p(oedge, ctypes.c_int32), _p(img, ctypes.c_double), _p(init, ctypes.c_double)))
        d.update(tiles=tiles, slot_ptr=slot_ptr, cnode=cnode, pitem=pitem, oedge=oedge[:d["E"]], cpt_img=img[:d["cpt_doubles"]], npi_init=init)
        return d

    def small_plan(self):
        """The plan of the one-workgroup path for small networks (bn_small_plan_get), or None when the network is not
        eligible: dict(n, N, M, S, T, TT, CL, waves, re, rb, rc, mmax, ent [re*nt, 2], ent_cpt, term, clist,
        bslot [rb*nt, 4], cslot [rc*nt, 4], npi_init)."""
        L = _lib.lib()
        if not self.info("small_eligible"):
            return None
        dims = np.zeros(12, dtype=np.int32)
        _lib.check(L.bn_small_plan_get(self._h, _p(dims, ctypes.c_int32), None, None, None, None, None, None, None))
        names = ["n", "N", "M", "S", "T", "TT", "CL", "waves", "re", "rb", "rc", "mmax"]
        d = {k: int(v) for k, v in zip(names, dims)}
        nt = 64 * d["waves"]
        u32 = ctypes.POINTER(ctypes.c_uint32)
        ent = np.zeros((d["re"] * nt, 2), dtype=np.uint32)
        ent_cpt = np.zeros(d["re"] * nt, dtype=np.float64)
        term = np.zeros(max(d["TT"], 1), dtype=np.uint32)
        clist = np.zeros(max(d["CL"], 1), dtype=np.uint16)
        bslot = np.zeros((d["rb"] * nt, 4), dtype=np.uint32)
        cslot = np.zeros((d["rc"] * nt, 4), dtype=np.uint32)
        init = np.zeros(d["N"], dtype=np.float64)
        _lib.check(L.bn_small_plan_get(self._h, None, ent.ctypes.data_as(u32), _p(ent_cpt, ctypes.c_double), term.ctypes.data_as(u32),
                                       clist.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), bslot.ctypes.data_as(u32),
                                       cslot.ctypes.data_as(u32), _p(init, ctypes.c_double)))
        d.update(ent=ent, ent_cpt=ent_cpt, term=term[:d["TT"]], clist=clist[:d["CL"]], bslot=bslot, cslot=cslot, npi_init=init)
        return d

    # single steps (tests): begin / sweep without exchange / finish
    def step_begin(self):
        _lib.check(_lib.lib().bn_bp_step_begin(self._h))

    def step_sweep(self, sweep: int, eps: float, part: int = 0):
        """part 0: the whole sweep; 1: interior tiles; 2: tiles touching a cut edge + bookkeeping."""
        _lib.check(_lib.lib().bn_bp_step_sweep_part(self._h, sweep, float(eps), part))

    def step_finish(self, launched: int, final: bool, eps: float):
        done, sw, res = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_double(0.0)
        _lib.check(_lib.lib().bn_bp_step_finish(self._h, launched, 1 if final else 0, float(eps), ctypes.byref(done),
                                                ctypes.byref(sw), ctypes.byref(res)))
        return done.value, sw.value, res.value

    def edge_refs(self):
        pi = np.zeros(max(self.model.n_edges, 1), dtype=np.int32)
        lam = np.zeros(max(self.model.n_edges, 1), dtype=np.int32)
        _lib.check(_lib.lib().bn_layout_edge_refs(self._h, _p(pi, ctypes.c_int32), _p(lam, ctypes.c_int32)))
        return pi[:self.model.n_edges], lam[:self.model.n_edges]

    # ---- likelihood weighting ----------------------------------------------------------
    def lw_run(self, ev_state, n_samples: int, seed: int, sample_begin: int = 0) -> np.ndarray:
        """Un-normalised weighted histogram [sum k] of samples [sample_begin, +n_samples)."""
        ev_state = np.asarray(ev_state, dtype=np.int32)
        nodes = np.ascontiguousarray(np.nonzero(ev_state >= 0)[0], dtype=np.int32)
        states = np.ascontiguousarray(ev_state[nodes], dtype=np.int32)
        hist = np.zeros(self._nbel, dtype=np.float64)
        _lib.check(_lib.lib().bn_lw_run(self._h, nodes.size, _p(nodes, ctypes.c_int32), _p(states, ctypes.c_int32),
                                        ctypes.c_uint64(sample_begin), ctypes.c_uint64(n_samples),
                                        ctypes.c_uint64(seed), _p(hist, ctypes.c_double)))
        return hist

    def lw_run_allreduce(self, ev_state, n_samples_total: int, seed: int, sample_begin: int = 0) -> np.ndarray:
        """All ranks of the communicator share the sample range; returns the summed histogram."""
        ev_state = np.asarray(ev_state, dtype=np.int32)
        nodes = np.ascontiguousarray(np.nonzero(ev_state >= 0)[0], dtype=np.int32)
        states = np.ascontiguousarray(ev_state[nodes], dtype=np.int32)
        hist = np.zeros(self._nbel, dtype=np.float64)
        _lib.check(_lib.lib().bn_lw_run_allreduce(self._h, nodes.size, _p(nodes, ctypes.c_int32),
                                                  _p(states, ctypes.c_int32), ctypes.c_uint64(sample_begin),
                                                  ctypes.c_uint64(n_samples_total), ctypes.c_uint64(seed),
                                                  _p(hist, ctypes.c_double)))
        return hist

    def rs_run(self, ev_state, n_accept: int, seed: int, max_draw: int = 1 << 34, sample_begin: int = 0):
        """Rejection sampling: (state counts of the first n_accept accepted samples, drawn, accepted)."""
        ev_state = np.asarray(ev_state, dtype=np.int32)
        nodes = np.ascontiguousarray(np.nonzero(ev_state >= 0)[0], dtype=np.int32)
        states = np.ascontiguousarray(ev_state[nodes], dtype=np.int32)
        counts = np.zeros(self._nbel, dtype=np.float64)
        drawn, acc = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _lib.check(_lib.lib().bn_rs_run(self._h, nodes.size, _p(nodes, ctypes.c_int32), _p(states, ctypes.c_int32),
                                        ctypes.c_uint64(sample_begin), ctypes.c_uint64(n_accept),
                                        ctypes.c_uint64(max_draw), ctypes.c_uint64(seed), _p(counts, ctypes.c_double),
                                        ctypes.byref(drawn), ctypes.byref(acc)))
        return counts, drawn.value, acc.value

    def fit_cpt(self, patterns, counts) -> np.ndarray:
        """fit_cpt() on this engine's structure."""
        return fit_cpt(self.model, patterns, counts)

    def lw_states(self, n: int):
        states = np.zeros((n, self.model.n), dtype=np.uint8)
        weights = np.zeros(n, dtype=np.float64)
        _lib.check(_lib.lib().bn_lw_states(self._h, ctypes.c_uint64(n), _p(states, ctypes.c_uint8),
                                           _p(weights, ctypes.c_double)))
        return states, weights

    # ---- layout ------------------------------------------------------------------------
    def layout(self) -> dict:
        li = _lib.LayoutInfo()
        _lib.check(_lib.lib().bn_layout_get(self._h, ctypes.byref(li)))
        return {f: getattr(li, f) for f, _ in li._fields_}

    def layout_classes(self):
        out = []
        for c in range(self.layout()["n_classes"]):
            v = [ctypes.c_int32() for _ in range(5)]
            _lib.check(_lib.lib().bn_layout_class(self._h, c, *[ctypes.byref(x) for x in v]))
            out.append(dict(zip(["kv", "m", "lanes_per_node", "variant", "n_nodes"], [x.value for x in v])))
        return out

    def node_tiles(self) -> np.ndarray:
        t = np.zeros(max(self.model.n, 1), dtype=np.int32)
        _lib.check(_lib.lib().bn_layout_node_tiles(self._h, _p(t, ctypes.c_int32)))
        return t[:self.model.n]

    def node_slots(self) -> np.ndarray:
        s = np.zeros(max(self.model.n, 1), dtype=np.int32)
        _lib.check(_lib.lib().bn_layout_node_slots(self._h, _p(s, ctypes.c_int32)))
        return s[:self.model.n]


def fit_cpt(model: FlatModel, patterns, counts, device: int = _lib.BN_DEVICE_CURRENT) -> np.ndarray:
    """sampler::make_cpt (reference sampler.hpp:81-163) on the GPU: flat CPTs of `model`'s structure
    fitted to a pattern table (patterns [P][n] uint8 states, counts [P]); model.cpt is not read."""
    patterns = np.ascontiguousarray(patterns, dtype=np.uint8).reshape(-1, max(model.n, 1))
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    if counts.shape[0] != patterns.shape[0]:
        raise ValueError("one count per pattern")
    out = np.zeros(int(model.cpt_off[-1]), dtype=np.float64)
    d = _lib.ModelDesc(model.n, _p(model.k, ctypes.c_int32), _p(model.in_ptr, ctypes.c_int32),
                       _p(model.in_idx, ctypes.c_int32), _p(model.cpt_off, ctypes.c_int64), None, device, 0)
    _lib.check(_lib.lib().bn_fit_cpt(ctypes.byref(d), patterns.shape[0], _p(patterns, ctypes.c_uint8),
                                     _p(counts, ctypes.c_uint64), _p(out, ctypes.c_double)))
    return out


class Sampler:
    """Mirror of bn::sampler (reference bayesian/sampler.hpp:17-215): a table of joint patterns with
    occurrence counts, and make_cpt() fitting every CPT of a structure to it on the GPU."""

    def __init__(self, filename: str = ""):
        self._filename, self._table, self._size = filename, {}, 0

    def filename(self) -> str:
        return self._filename

    def set_filename(self, filename: str) -> None:  # :172-177 resets the table
        self._filename, self._table, self._size = filename, {}, 0

    def table(self) -> dict:
        return dict(self._table)

    def sampling_size(self) -> int:
        return self._size

    def load_sample(self, arg) -> bool:
        """dict {pattern tuple (state per node id): count} (:29-37) or a node order for the
        whitespace-separated file "count s_0 s_1 ..." set by set_filename (:42-77; False if unreadable)."""
        if isinstance(arg, dict):
            self._table = {tuple(int(x) for x in key): int(c) for key, c in arg.items()}
            self._size = sum(self._table.values())
            return True
        order = [int(v) for v in arg]
        try:
            fh = open(self._filename)
        except OSError:
            return False
        table, size = {}, 0
        with fh:
            for line in fh:
                tok = line.split()
                if not tok:
                    continue
                cnt = int(tok[0])
                pat = [0] * len(order)
                for pos, v in enumerate(order):
                    pat[v] = int(tok[1 + pos])
                key = tuple(pat)
                table[key] = table.get(key, 0) + cnt
                size += cnt
        self._table, self._size = table, size
        return True

    def make_cpt(self, model: FlatModel) -> bool:
        """Overwrites model.cpt with the fitted CPTs (:81-163); False when no sample is loaded (:83)."""
        if self._size == 0:
            return False
        pats = np.array(list(self._table.keys()), dtype=np.uint8).reshape(len(self._table), model.n)
        cnts = np.array(list(self._table.values()), dtype=np.uint64)
        model.cpt[:] = fit_cpt(model, pats, cnts)
        return True


def debug_allgather(engines, sweep: int) -> None:
    """Emulated exchange between shard engines living on one device (bn_debug_allgather)."""
    arr = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    _lib.check(_lib.lib().bn_debug_allgather(arr, len(engines), sweep))


def run_shards_on_one_device(engines, evidence, eps: float, max_sweeps: int = 0, overlapped: bool = True):
    """Drive n shard engines on ONE GPU through the step API with the emulated all-gather:
    the same kernels, layout and stopping logic as the RCCL path, minus the collective itself.
    overlapped: the launch order of the engine's overlapped run -- the interior tiles of sweep s are
    launched (and here even finished) BEFORE the exchange of sweep s-1 is applied, the tiles that touch
    a cut edge after it; results must not depend on it (interior tiles read nothing the exchange delivers)."""
    for e in engines:
        e.bp_set_evidence(evidence)
        e.step_begin()
    launched, batch = 0, 4
    while True:
        if max_sweeps > 0:
            batch = min(batch, max_sweeps - launched)
        for i in range(batch):
            s = launched + i
            if overlapped:
                for e in engines:
                    e.step_sweep(s, eps, part=1)
                if s > 0:
                    debug_allgather(engines, s - 1)
                for e in engines:
                    e.step_sweep(s, eps, part=2)
            else:
                for e in engines:
                    e.step_sweep(s, eps)
                debug_allgather(engines, s)
        if overlapped and batch > 0:
            debug_allgather(engines, launched + batch - 1)
        launched += batch
        outs = [e.step_finish(launched, max_sweeps > 0 and launched >= max_sweeps, eps) for e in engines]
        if len({o[0] != 0 for o in outs}) != 1 or len({o[1] for o in outs if o[0]}) > 1:
            raise RuntimeError(f"shards disagree on convergence: {outs}")
        if outs[0][0] != 0:
            return {"sweeps": outs[0][1], "residual": outs[0][2]}


def _split(model: FlatModel, flat: np.ndarray):
    off = model.node_off
    return [flat[off[v]:off[v + 1]] for v in range(model.n)]


class BeliefPropagation:
    """``bn::inference::belief_propagation``: ``bp = BeliefPropagation(model); marg = bp(evidence, 0.001)``.

    Returns a list indexed by node (position in ``vertex_list()``) of 1 x k arrays, the python
    spelling of ``unordered_map<vertex_type, matrix_type>`` (belief_propagation.hpp:14)."""

    def __init__(self, model: FlatModel, device: int = _lib.BN_DEVICE_CURRENT):
        self.engine = Engine(model, device)
        self.model = model
        self.last = None

    def __call__(self, precondition=None, epsilon: float = 0.001):
        if isinstance(precondition, (int, float)) and not isinstance(precondition, bool):
            precondition, epsilon = None, float(precondition)  # the by-pass overload bp(epsilon), :24-28
        if isinstance(precondition, dict):
            precondition = Evidence.from_dict(self.model, precondition)
        self.last = self.engine.bp_run_view(precondition, epsilon)
        self.last["beliefs"] = self.last["beliefs"].copy()  # out of the engine's pinned buffer: the caller owns the result
        return _split(self.model, self.last["beliefs"])

    def run_batch(self, preconditions, epsilon: float = 0.001):
        """Not in the reference (one query per call): several evidence sets in one call; entry q is exactly what
        ``self(preconditions[q], epsilon)`` returns.  Lists longer than BN_MAX_BATCH_SETS go in slices."""
        evs = [Evidence.from_dict(self.model, p) if isinstance(p, dict) else p for p in preconditions]
        out = []
        for b in range(0, len(evs), _lib.BN_MAX_BATCH_SETS):
            self.last = self.engine.bp_run_batch(evs[b:b + _lib.BN_MAX_BATCH_SETS], epsilon)
            out += [_split(self.model, bel) for bel in self.last["beliefs"]]
        return out


class LikelihoodWeighting:
    """``bn::inference::likelihood_weighting``: ``lw = LikelihoodWeighting(model); marg = lw({node: state}, 10000)``."""

    def __init__(self, model: FlatModel, device: int = _lib.BN_DEVICE_CURRENT, seed: int = 0x5EED):
        self.engine = Engine(model, device)
        self.model = model
        self.seed = seed
        self._next = 0
        self.last_units = 0

    def __call__(self, evidence=None, sample_num: int = 10000):
        ev_state = np.full(self.model.n, -1, dtype=np.int32)
        for v, s in (evidence or {}).items():
            ev_state[v] = s
        hist = self.engine.lw_run(ev_state, sample_num, self.seed, self._next)
        self._next += sample_num  # successive calls continue the stream like the reference's engine
        return _split(self.model, normalize_histogram(self.model, hist))


    def make_samples(self, evidence=None, unit_size: int = 1000000, epsilon: float = 0.001):
        """``likelihood_weighting::make_samples`` (likelihood_weighting.hpp:62-117): units of
        ``unit_size`` weighted samples are drawn until no normalised marginal moved by ``epsilon`` or
        more between two consecutive units.  Returns ``(patterns, marginals)``: the occurrence count of
        every complete joint pattern over all units (dict: state tuple -> count, the python spelling
        of ``unordered_map<condition_t, size_t>``) and the marginals of the last unit.  The patterns are
        read back from the GPU's state matrix piece by piece; ``self.last_units`` = units executed."""
        ev_state = np.full(self.model.n, -1, dtype=np.int32)
        for v, s in (evidence or {}).items():
            ev_state[v] = s
        n, hn = self.model.n, int(self.model.k.sum())
        w_list, prob = np.zeros(hn), np.zeros(hn)
        table: dict = {}
        piece_cap = max(1, min(unit_size, (1 << 28) // max(n, 1), 1 << 20))
        self.last_units = 0
        while True:
            done = 0
            while done < unit_size:  # one unit (:82-99), in pieces the device keeps in one state matrix
                piece = min(unit_size - done, piece_cap)
                w_list += self.engine.lw_run(ev_state, piece, self.seed, self._next)
                self._next += piece
                states, _ = self.engine.lw_states(piece)
                pats, cnts = np.unique(states, axis=0, return_counts=True)
                for pat, c in zip(pats, cnts):
                    key = pat.tobytes()
                    table[key] = table.get(key, 0) + int(c)
                done += piece
            self.last_units += 1
            nxt = normalize_histogram(self.model, w_list)  # :101-112
            diff = max(np.finfo(np.float64).tiny, float(np.abs(prob - nxt).max())) if hn else np.finfo(np.float64).tiny
            prob = nxt
            if diff < epsilon:
                break
        patterns = {tuple(np.frombuffer(key, dtype=np.uint8).tolist()): c for key, c in table.items()}
        return patterns, _split(self.model, prob)


def normalize_histogram(model: FlatModel, hist: np.ndarray) -> np.ndarray:
    """likelihood_weighting.hpp:197-221: divide by the sum; uniform when the sum < 1e-20."""
    out = hist.astype(np.float64).copy()
    off = model.node_off
    for v in range(model.n):
        h = out[off[v]:off[v + 1]]
        s = 0.0
        for x in h:
            s += x
        if s < 1.0e-20:
            h[:] = 1.0 / h.size
        else:
            h /= s
    return out


class RejectionSampling:
    """``bn::inference::rejection_sampling`` (reference rejection_sampling.hpp:13-62):
    ``rs = RejectionSampling(model); marg = rs({node: state}, 10000)``."""

    def __init__(self, model: FlatModel, device: int = _lib.BN_DEVICE_CURRENT, seed: int = 0x5EED, max_draws: int = 1 << 34):
        self.engine = Engine(model, device)
        self.model, self.seed, self.max_draws, self._next = model, seed, max_draws, 0
        self.last_drawn = 0

    def __call__(self, condition=None, generate_sample_num: int = 10000):
        ev_state = np.full(self.model.n, -1, dtype=np.int32)
        for v, s in (condition or {}).items():
            ev_state[v] = s
        counts, drawn, acc = self.engine.rs_run(ev_state, generate_sample_num, self.seed, self.max_draws, self._next)
        self._next += drawn
        self.last_drawn = drawn
        if acc < generate_sample_num:
            raise RuntimeError("rejection sampling: condition too unlikely, gave up after max_draws samples")
        return _split(self.model, counts / acc)

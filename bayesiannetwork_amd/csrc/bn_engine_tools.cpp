// bn_engine_tools.cpp -- plan and layout introspection (what the CPU emulators and tests read), bn_reload_cpt, and the entry points of
// likelihood weighting, rejection sampling and the CPT fit.
#include "bn_engine_internal.hpp"

// ---- layout introspection ---------------------------------------------------------------------
// The plan of the one-workgroup path (bn_small.hpp; tests emulate the kernel on it).  dims_out[12] = n, N, M, S, T, TT, CL,
// waves, re, rb, rc, mmax; the arrays (any may be null) are sized from those: ent [re * 64 waves][2], ent_cpt [re * 64 waves],
// term [TT], clist [CL], bslot / cslot [rb | rc * 64 waves][4], npi_init [N].  BN_ERR_STATE when the network is not eligible.
static int small_plan_copy(const SmallPlan& sp, const SmallPlan& tables, int32_t* dims_out, uint32_t* ent, double* ent_cpt, uint32_t* term,
                           uint16_t* clist, uint32_t* bslot, uint32_t* cslot, double* npi_init);
extern "C" int bn_small_plan_get(bn_engine* e, int32_t* dims_out, uint32_t* ent, double* ent_cpt, uint32_t* term, uint16_t* clist,
                                 uint32_t* bslot, uint32_t* cslot, double* npi_init) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    const SmallPlan& sp = e->small;
    if (!sp.ok) return fail(BN_ERR_STATE, "not eligible for the one-workgroup path: " + (sp.why.empty() ? std::string("disabled") : sp.why));
    return small_plan_copy(sp, sp, dims_out, ent, ent_cpt, term, clist, bslot, cslot, npi_init);
}
// ... of part `part` of the plan that spreads a mid-size network over several workgroups (bn_get_info "mid_parts"); the same
// layout (message / node-vector indices global, staging places the part's own); dims_out[12..13] = the part's node range
extern "C" int bn_mid_plan_get(bn_engine* e, int32_t part, int32_t* dims_out, uint32_t* ent, double* ent_cpt, uint32_t* term, uint16_t* clist,
                               uint32_t* bslot, uint32_t* cslot, double* npi_init) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (!e->mid.ok) return fail(BN_ERR_STATE, "not eligible for the mid-size path: " + (e->mid.why.empty() ? std::string("not needed or disabled") : e->mid.why));
    if (part < 0 || part >= int32_t(e->mid.parts.size())) return fail(BN_ERR_ARG, "part index out of range");
    const SmallPlan& sp = e->mid.parts[part];
    if (dims_out) { dims_out[12] = sp.v0; dims_out[13] = sp.v1; }
    return small_plan_copy(sp, e->mid.parts[0], dims_out, ent, ent_cpt, term, clist, bslot, cslot, npi_init);
}
static int small_plan_copy(const SmallPlan& sp, const SmallPlan& tables, int32_t* dims_out, uint32_t* ent, double* ent_cpt, uint32_t* term,
                           uint16_t* clist, uint32_t* bslot, uint32_t* cslot, double* npi_init) {
    if (dims_out) {
        const int32_t d[12] = {sp.n, sp.N, sp.M, sp.S, sp.T, sp.TT, sp.CL, sp.waves, sp.re, sp.rb, sp.rc, sp.mmax};
        std::copy(d, d + 12, dims_out);
    }
    if (ent) std::memcpy(ent, sp.ent.data(), sp.ent.size() * sizeof(SmallEntry));
    if (ent_cpt) std::copy(sp.ent_cpt.begin(), sp.ent_cpt.end(), ent_cpt);
    if (term) std::copy(sp.term.begin(), sp.term.begin() + sp.TT, term);
    if (clist) std::copy(sp.clist.begin(), sp.clist.begin() + sp.CL, clist);
    if (bslot) std::memcpy(bslot, sp.bslot.data(), sp.bslot.size() * sizeof(SmallSlot));
    if (cslot) std::memcpy(cslot, sp.cslot.data(), sp.cslot.size() * sizeof(SmallSlot));
    if (npi_init) std::copy(tables.npi_init.begin(), tables.npi_init.end(), npi_init);
    return BN_OK;
}

// The plan of the register-resident DAG path (bn_dag.hpp; tests emulate the kernel on it).
extern "C" int bn_dag_plan_get(bn_engine* e, int32_t* dims_out, int32_t* tiles, int32_t* slot_ptr, int32_t* cnode, int32_t* pitem,
                               int32_t* oedge, double* cpt_img, double* npi_init) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (int rc = ensure_dag(e)) return rc;   // (bn_create keeps the light plan until the path or its plan is asked for)
    const DagPlan& dp = e->dag;
    if (dims_out) {
        const int32_t d[8] = {dp.n, dp.E, int32_t(dp.tiles.size()), dp.blocks, dp.stream ? 1 : 0, dp.n_child_tiles, dp.n_parent_tiles,
                              int32_t(dp.cpt_img.size())};
        std::copy(d, d + 8, dims_out);
    }
    if (tiles) std::memcpy(tiles, dp.tiles.data(), dp.tiles.size() * sizeof(DagTile));
    if (slot_ptr) std::copy(dp.slot_ptr.begin(), dp.slot_ptr.end(), slot_ptr);
    if (cnode) std::memcpy(cnode, dp.cnode.data(), dp.cnode.size() * sizeof(DagChildLane));
    if (pitem) std::memcpy(pitem, dp.pitem.data(), dp.pitem.size() * sizeof(DagParentLane));
    if (oedge) std::copy(dp.oedge.begin(), dp.oedge.end(), oedge);
    if (cpt_img) std::copy(dp.cpt_img.begin(), dp.cpt_img.end(), cpt_img);
    if (npi_init) std::copy(dp.npi_init.begin(), dp.npi_init.end(), npi_init);
    return BN_OK;
}

// New CPT values on an unchanged structure (belief_propagation.hpp:61,186,252: the reference reads node->cpt on every call, so a
// table edited or re-fitted after the functor was built IS seen there; here the tables are device images made at bn_create).
// Re-derives every image that holds CPT values -- the lane-striped tile image, the entry tables of the item kernels, the
// register image of the DAG path, the initial pi(v) of the roots -- from the new flat array and copies them over the old ones;
// the sampler state is rebuilt at its next call.  No allocation changes size.
template <class T, class A>
static int reupload(T* dst, const std::vector<T, A>& src, size_t expect, hipStream_t s, const char* what) {
    if (src.size() != expect) return fail(BN_ERR_STATE, std::string("bn_reload_cpt: the ") + what + " changed size (structure changed?)");
    if (!src.empty()) HIPCHK(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return BN_OK;
}
extern "C" int bn_reload_cpt(bn_engine* e, const double* cpt, int64_t n_entries) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: an earlier bn_reload_cpt failed while uploading (destroy it and create a new one)");
    Plan& p = e->plan;
    const int64_t want = p.n > 0 ? p.cpt_off[p.n] : 0;
    if (n_entries != want) return fail(BN_ERR_ARG, "bn_reload_cpt: " + std::to_string(n_entries) + " entries given, the model has " + std::to_string(want));
    if (want > 0 && !cpt) return fail(BN_ERR_ARG, "null cpt");
    // Two phases.  (1) every host plan that holds CPT values is rebuilt from the new array into TEMPORARIES and checked against
    // the one in use; a mismatch returns with the engine exactly as it was.  (2) the temporaries are swapped in and the device
    // images overwritten; a HIP failure there leaves device images of mixed age, so the engine is marked unusable.
    BigVec old_flat;
    bool swapped_flat = false;
    try {
        old_flat.assign(cpt, cpt + want);
        old_flat.swap(p.cpt_flat);   // the planners read p.cpt_flat; old_flat now holds the values in force
        swapped_flat = true;
        auto refuse = [&](const char* what) {
            p.cpt_flat.swap(old_flat);
            return fail(BN_ERR_STATE, std::string("bn_reload_cpt: ") + what);
        };
        SmallPlan n_small;
        MidPlan n_mid;
        DagPlan n_dag;
        if (e->small.ok) {
            build_small_plan(p, n_small);
            if (!n_small.ok || n_small.ent_cpt.size() != e->small.ent_cpt.size() || n_small.npi_init.size() != e->small.npi_init.size())
                return refuse("the one-workgroup plan changed");
        }
        if (e->mid.ok) {
            build_mid_plan(p, n_mid);
            bool same = n_mid.ok && n_mid.parts.size() == e->mid.parts.size();
            for (size_t q = 0; same && q < n_mid.parts.size(); ++q) same = n_mid.parts[q].ent_cpt.size() == e->mid.parts[q].ent_cpt.size();
            same = same && (n_mid.parts.empty() || n_mid.parts[0].npi_init.size() == e->mid.parts[0].npi_init.size());
            if (!same) return refuse("the plan of the several-workgroup path changed");
        }
        if (e->dag.ok) {
            build_dag_plan(p, e->dag_cap, n_dag, e->dag.light);   // (the same cap gives the same plan; only the values differ.  A plan that is still light stays light)
            bool same = n_dag.ok && n_dag.cpt_img.size() == e->dag.cpt_img.size() && n_dag.blocks == e->dag.blocks &&
                        n_dag.npi_init.size() == e->dag.npi_init.size() && n_dag.tiles.size() == e->dag.tiles.size() && n_dag.slot_ptr == e->dag.slot_ptr;
            // the device tables (tile-major record numbers, eperm / nperm) were derived from the plan in use and stay: the new image must
            // belong to the very same tiles
            for (size_t t = 0; same && t < n_dag.tiles.size(); ++t) {
                const DagTile &x = n_dag.tiles[t], &y = e->dag.tiles[t];
                same = x.kind == y.kind && x.n_active == y.n_active && x.lane_base == y.lane_base && x.cpt_base == y.cpt_base;
            }
            if (!same) return refuse("the plan of the register-resident DAG path changed");
        }
        if (!e->host_only) {
            ON_DEVICE(e);
            if (hipStreamSynchronize(e->stream) != hipSuccess) return refuse("the engine's stream reports an error");   // nothing of the old tables is in use any more
            stripe_cpt(p, cpt);
            if (p.cpt_striped.size() != size_t(p.cpt_doubles)) { BigVec().swap(p.cpt_striped); return refuse("the tile image changed size"); }
        }
        // ---- commit
        if (e->small.ok) e->small = std::move(n_small);
        if (e->mid.ok) e->mid = std::move(n_mid);
        if (e->dag.ok) e->dag = std::move(n_dag);
        if (e->dense) { free_engine(e->dense); e->dense = nullptr; e->batch_on_dense = false; }   // (rebuilt from the new tables on demand)
        if (e->host_only) return BN_OK;
        e->poisoned = true;   // until every image has arrived
        ON_DEVICE(e);
        hipStream_t s = e->stream;
        int rc;
        if ((rc = reupload(e->d_cpt, p.cpt_striped, size_t(p.cpt_doubles), s, "tile image"))) return rc;
        if (e->small_ok) {
            if ((rc = reupload(e->d_s_cpt, e->small.ent_cpt, e->small.ent_cpt.size(), s, "entry table"))) return rc;
            if ((rc = reupload(e->d_s_init, e->small.npi_init, e->small.npi_init.size(), s, "initial pi"))) return rc;
        }
        if (e->mid_ok) {
            std::vector<double> all;
            for (const SmallPlan& sp : e->mid.parts) all.insert(all.end(), sp.ent_cpt.begin(), sp.ent_cpt.end());
            if ((rc = reupload(e->d_m_cpt, all, all.size(), s, "entry tables"))) return rc;
            HIPCHK(hipStreamSynchronize(s));   // `all` is a local
            if ((rc = reupload(e->d_m_init, e->mid.parts[0].npi_init, e->mid.parts[0].npi_init.size(), s, "initial pi"))) return rc;
        }
        if (e->dag_ready) {
            if ((rc = reupload(e->d_g_cpt, e->dag.cpt_img, e->dag.cpt_img.size(), s, "register image"))) return rc;
            if ((rc = reupload(e->d_g_init, e->dag.npi_init, e->dag.npi_init.size(), s, "initial pi"))) return rc;
        }
        HIPCHK(hipStreamSynchronize(s));
        BigVec().swap(p.cpt_striped);
        lw_free(e->lw);   // the sampler uploads its copy of the tables at its next call
        e->poisoned = false;
    } catch (const std::bad_alloc&) {
        if (swapped_flat && !e->poisoned) p.cpt_flat.swap(old_flat);   // phase 1: nothing was committed
        return fail(BN_ERR_ALLOC, "out of host memory in bn_reload_cpt");
    }
    return BN_OK;
}

extern "C" int bn_layout_get(bn_engine* e, bn_layout_info* o) {
    if (!e || !o) return fail(BN_ERR_ARG, "null argument");
    const Plan& p = e->plan;
    o->n_nodes = p.n;
    o->n_edges = int32_t(p.E);
    o->n_classes = int32_t(p.classes.size());
    o->n_tiles = int32_t(p.tiles.size());
    o->lanes_per_node_max = p.g_max;
    o->cpt_doubles = p.cpt_doubles;
    o->rec_doubles = p.rec_doubles;
    o->node_doubles = p.node_doubles;
    o->algorithmic_bytes_per_sweep = p.algorithmic_bytes;
    o->layout_bytes_per_sweep = p.layout_bytes;
    o->messages_per_sweep = p.messages_per_sweep;
    o->rank = p.rank;
    o->nranks = p.nranks;
    o->n_owned = p.n_owned;
    o->n_interior_tiles = p.n_interior_tiles;
    o->n_cut_edges = p.n_cut_edges;
    o->segment_bytes = p.seg_d2 * 16;
    o->segment_used_bytes = p.seg_used_d2.empty() ? 0 : p.seg_used_d2[p.rank] * 16;
    o->exchange_base = p.g_base;
    return BN_OK;
}

extern "C" int bn_layout_node_tiles(bn_engine* e, int32_t* tiles_out) {
    if (!e || !tiles_out) return fail(BN_ERR_ARG, "null argument");
    std::copy(e->plan.node_tile.begin(), e->plan.node_tile.end(), tiles_out);
    return BN_OK;
}

extern "C" int bn_layout_node_slots(bn_engine* e, int32_t* slots_out) {
    if (!e || !slots_out) return fail(BN_ERR_ARG, "null argument");
    std::copy(e->plan.node_slot.begin(), e->plan.node_slot.end(), slots_out);
    return BN_OK;
}

extern "C" int bn_layout_edge_refs(bn_engine* e, int32_t* pi_out, int32_t* lam_out) {
    if (!e || !pi_out || !lam_out) return fail(BN_ERR_ARG, "null argument");
    for (int64_t i = 0; i < e->plan.E; ++i) { pi_out[i] = e->plan.edge_ref[i].pi; lam_out[i] = e->plan.edge_ref[i].lam; }
    return BN_OK;
}

extern "C" int bn_layout_class(bn_engine* e, int32_t cls, int32_t* kv, int32_t* m, int32_t* lanes_per_node,
                               int32_t* variant, int32_t* n_nodes) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (cls < 0 || cls >= int32_t(e->plan.classes.size())) return fail(BN_ERR_ARG, "class index out of range");
    const ClassDesc& c = e->plan.classes[cls];
    if (kv) *kv = c.kv;
    if (m) *m = c.m;
    if (lanes_per_node) *lanes_per_node = c.G;
    if (variant) *variant = c.variant;
    if (n_nodes) *n_nodes = c.n_nodes;
    return BN_OK;
}

// ---- likelihood weighting -------------------------------------------------------------------------
extern "C" int bn_lw_run(bn_engine* e, int32_t ne, const int32_t* ev_node, const int32_t* ev_state,
                         uint64_t sample_begin, uint64_t n_samples, uint64_t seed, double* hist_out) {
    if (!e || !hist_out) return fail(BN_ERR_ARG, "null argument");
    if (e->host_only) return fail(BN_ERR_STATE, "engine was created with BN_DEVICE_HOST_ONLY: no GPU, no compute");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: bn_reload_cpt failed while uploading (destroy it and create a new one)");
    if (ne < 0 || (ne > 0 && (!ev_node || !ev_state))) return fail(BN_ERR_ARG, "bad evidence arguments");
    ON_DEVICE(e);
    std::string err;
    int rc = lw_run(e->lw, e->plan, e->stream, ne, ev_node, ev_state, sample_begin, n_samples, seed, hist_out, err);
    if (rc) return fail(rc, err);
    return BN_OK;
}

// Likelihood weighting over every rank of the communicator: the sample range is split evenly,
// each GPU draws its share (disjoint sample ids = disjoint streams), ONE RCCL all-reduce sums the histograms.
extern "C" int bn_lw_run_allreduce(bn_engine* e, int32_t ne, const int32_t* ev_node, const int32_t* ev_state,
                                   uint64_t sample_begin, uint64_t n_samples_total, uint64_t seed, double* hist_out) {
    if (!e || !hist_out) return fail(BN_ERR_ARG, "null argument");
    if (e->host_only) return fail(BN_ERR_STATE, "engine was created with BN_DEVICE_HOST_ONLY: no GPU, no compute");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: bn_reload_cpt failed while uploading (destroy it and create a new one)");
    if (ne < 0 || (ne > 0 && (!ev_node || !ev_state))) return fail(BN_ERR_ARG, "bad evidence arguments");
    if (!e->comm) return fail(BN_ERR_COMM, "call bn_comm_init first");
    ON_DEVICE(e);
    const uint64_t P = uint64_t(e->plan.nranks), r = uint64_t(e->plan.rank);
    const uint64_t lo = n_samples_total * r / P, hi = n_samples_total * (r + 1) / P;
    std::string err;
    int rc = lw_run(e->lw, e->plan, e->stream, ne, ev_node, ev_state, sample_begin + lo, hi - lo, seed, nullptr, err);
    if (rc) return fail(rc, err);
    const size_t hist_n = size_t(e->plan.node_off[e->plan.n]);
    ncclResult_t nr = g_rccl.AllReduce(e->lw.d_hist, e->lw.d_hist, hist_n, ncclDouble, ncclSum, e->comm, e->stream);
    if (nr != ncclSuccess) return fail(BN_ERR_COMM, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(nr));
    HIPCHK(hipMemcpyAsync(hist_out, e->lw.d_hist, hist_n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return BN_OK;
}

// Rejection (logic) sampling, reference rejection_sampling.hpp:33-167.
extern "C" int bn_rs_run(bn_engine* e, int32_t ne, const int32_t* ev_node, const int32_t* ev_state,
                         uint64_t sample_begin, uint64_t n_accept, uint64_t max_draw, uint64_t seed, double* counts_out,
                         uint64_t* drawn_out, uint64_t* accepted_out) {
    if (!e || !counts_out) return fail(BN_ERR_ARG, "null argument");
    if (e->host_only) return fail(BN_ERR_STATE, "engine was created with BN_DEVICE_HOST_ONLY: no GPU, no compute");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: bn_reload_cpt failed while uploading (destroy it and create a new one)");
    if (ne < 0 || (ne > 0 && (!ev_node || !ev_state))) return fail(BN_ERR_ARG, "bad condition arguments");
    if (max_draw == 0) return fail(BN_ERR_ARG, "max_draw must be > 0 (the reference loops forever on impossible evidence)");
    ON_DEVICE(e);
    std::string err;
    int rc = rs_run(e->lw, e->plan, e->stream, ne, ev_node, ev_state, sample_begin, n_accept, max_draw, seed,
                    counts_out, drawn_out, accepted_out, err);
    if (rc) return fail(rc, err);
    return BN_OK;
}

// CPT fitting from a table of joint patterns, reference sampler.hpp:81-163 (sampler::make_cpt).
// Stateless: only the structure of `desc` is read (its cpt pointer is ignored and may be null).
extern "C" int bn_fit_cpt(const bn_model_desc* desc, int64_t n_patterns, const uint8_t* patterns,
                          const uint64_t* counts, double* cpt_out) {
    if (!desc || !cpt_out) return fail(BN_ERR_ARG, "null argument");
    if (desc->n_nodes < 0 || (desc->n_nodes > 0 && (!desc->k || !desc->in_ptr || !desc->cpt_off)))
        return fail(BN_ERR_ARG, "bad model structure");
    if (n_patterns < 0 || (n_patterns > 0 && (!patterns || !counts))) return fail(BN_ERR_ARG, "bad pattern table");
    const int32_t n = desc->n_nodes;
    std::vector<int32_t> k(desc->k, desc->k + n), in_ptr(desc->in_ptr, desc->in_ptr + n + 1);
    std::vector<int64_t> cpt_off(desc->cpt_off, desc->cpt_off + n + 1);
    if (in_ptr[0] != 0 || cpt_off[0] != 0) return fail(BN_ERR_ARG, "in_ptr / cpt_off must start at 0");
    for (int32_t v = 0; v < n; ++v) {
        if (k[v] < 1 || k[v] > 255) return fail(BN_ERR_ARG, "node arity must be in 1..255");
        if (in_ptr[v + 1] < in_ptr[v] || in_ptr[v + 1] - in_ptr[v] > BN_MAX_PARENTS) return fail(BN_ERR_ARG, "bad in_ptr");
    }
    if (in_ptr[n] > 0 && !desc->in_idx) return fail(BN_ERR_ARG, "bad model structure");
    std::vector<int32_t> in_idx(desc->in_idx, desc->in_idx + in_ptr[n]);
    for (int32_t v = 0; v < n; ++v) {
        int64_t rows = 1;
        for (int32_t e = in_ptr[v]; e < in_ptr[v + 1]; ++e) {
            if (in_idx[e] < 0 || in_idx[e] >= n || in_idx[e] == v) return fail(BN_ERR_ARG, "parent index out of range");
            rows *= k[in_idx[e]];
        }
        if (cpt_off[v + 1] - cpt_off[v] != rows * k[v]) return fail(BN_ERR_ARG, "cpt_off does not match the arities");
    }
    // sampler::make_cpt returns false on an empty table (:83); here that is an argument error
    uint64_t total = 0;
    for (int64_t i = 0; i < n_patterns; ++i) total += counts[i];
    if (total == 0) return fail(BN_ERR_ARG, "empty sample table (sampling_size() == 0)");
    // [node][pattern] image so that a wave reads contiguous bytes
    std::vector<uint8_t> tr(std::max<size_t>(size_t(n) * size_t(n_patterns), 1));
    for (int64_t i = 0; i < n_patterns; ++i)
        for (int32_t v = 0; v < n; ++v) {
            if (patterns[i * n + v] >= k[v]) return fail(BN_ERR_ARG, "pattern state out of range");
            tr[size_t(v) * n_patterns + i] = patterns[i * n + v];
        }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(BN_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (desc->device >= ndev) return fail(BN_ERR_ARG, "device ordinal out of range");
    if (desc->device >= 0) HIPCHK(hipSetDevice(desc->device));
    std::vector<int32_t> row_node;
    std::vector<int64_t> row_off;
    for (int32_t v = 0; v < n; ++v)
        for (int64_t o = cpt_off[v]; o < cpt_off[v + 1]; o += k[v]) { row_node.push_back(v); row_off.push_back(o); }
    const size_t entries = size_t(cpt_off[n]);
    hipStream_t s = nullptr;
    uint8_t* d_pat = nullptr; unsigned long long* d_w = nullptr; unsigned long long* d_cnt = nullptr;
    int32_t *d_k = nullptr, *d_ptr = nullptr, *d_idx = nullptr, *d_rn = nullptr;
    int64_t *d_off = nullptr, *d_ro = nullptr; double* d_out = nullptr;
    int rc = [&]() -> int {
        int r;
        HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        if ((r = upload(&d_pat, tr, s))) return r;
        std::vector<unsigned long long> w(counts, counts + n_patterns);
        if ((r = upload(&d_w, w, s))) return r;
        if ((r = upload(&d_k, k, s))) return r;
        if ((r = upload(&d_ptr, in_ptr, s))) return r;
        if ((r = upload(&d_idx, in_idx, s))) return r;
        if ((r = upload(&d_off, cpt_off, s))) return r;
        if ((r = upload(&d_rn, row_node, s))) return r;
        if ((r = upload(&d_ro, row_off, s))) return r;
        if ((r = dalloc(&d_cnt, entries))) return r;
        if ((r = dalloc(&d_out, entries))) return r;
        HIPCHK(hipMemsetAsync(d_cnt, 0, std::max<size_t>(entries, 1) * 8, s));
        FitArgs a{n, d_k, d_ptr, d_idx, d_off, n_patterns, d_pat, d_w, d_cnt, int64_t(row_node.size()), d_rn, d_ro, d_out};
        if (launch_fit(a, s)) return fail(BN_ERR_HIP, "fit kernel launch failed");
        if (entries) HIPCHK(hipMemcpyAsync(cpt_out, d_out, entries * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return BN_OK;
    }();
    void* ptrs[] = {d_pat, d_w, d_cnt, d_k, d_ptr, d_idx, d_rn, d_off, d_ro, d_out};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    if (s) (void)hipStreamDestroy(s);
    return rc;
}

extern "C" int bn_lw_states(bn_engine* e, uint64_t n, uint8_t* states_out, double* weights_out) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (e->host_only) return fail(BN_ERR_STATE, "host-only engine");
    ON_DEVICE(e);
    std::string err;
    int rc = lw_states(e->lw, e->plan, e->stream, n, states_out, weights_out, err);
    if (rc) return fail(rc, err);
    return BN_OK;
}

// The achievable-HBM yardstick (csrc/bn_stream.hip): SURVEY 8(d) -- "measure the achievable figure with a stream-triad kernel on the
// box and report against both".  Stateless; device = HIP ordinal or BN_DEVICE_CURRENT.
double bn_stream_measure(int mode, size_t bytes, int reps, double* check_out);
extern "C" int bn_debug_stream(int32_t device, int32_t mode, int64_t bytes, int32_t reps, double* gbs_out) {
    if (!gbs_out) return fail(BN_ERR_ARG, "null argument");
    if (mode < 0 || mode > 1) return fail(BN_ERR_ARG, "mode: 0 copy, 1 triad");
    if (bytes < (1 << 20) || bytes > (int64_t(16) << 30) || reps < 1 || reps > 100) return fail(BN_ERR_ARG, "bytes in 1 MiB..16 GiB, reps in 1..100");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(BN_ERR_NO_DEVICE, "no HIP device");
    DeviceGuard guard;
    if (device != BN_DEVICE_CURRENT) HIPCHK(guard.enter(device));
    double check = 0.0;
    const double g = bn_stream_measure(mode, size_t(bytes), reps, &check);
    if (g < 0) return fail(BN_ERR_HIP, std::string("bn_debug_stream: ") + hipGetErrorString(hipError_t(int(-g))));
    // copy of {1, 2} -> 3; triad {1, 2} + 0.5 * {2, 3} -> 5.5
    if (check != (mode == 0 ? 3.0 : 5.5)) return fail(BN_ERR_STATE, "bn_debug_stream: the kernel's output is wrong");
    *gbs_out = g;
    return BN_OK;
}

// bn_engine_batch.cpp -- several evidence sets on one network per call (bn_bp_set_evidence_batch / bn_bp_run_batch*): an extension beside
// the drop-in, whose API takes one query at a time.  Every set keeps the bits, the sweep count and the residual history of its single run.
#include "bn_engine_internal.hpp"

// ---- several evidence sets on one network (extension beside the drop-in: the reference's API takes one
// query at a time).  Resident-eligible networks run all sets in ONE launch that walks them round-robin
// (bn_resident.hip): one resident CPT serves every set and each set's barrier completes while the others
// compute.  Other networks run the sets one after another through the single-query path.  Either way
// every set's results are bit-identical to running it alone.
static int batch_reserve(bn_engine* e, int32_t n_sets) {
    bn_engine::Batch& bt = e->batch;
    if (n_sets <= bt.cap_sets) return BN_OK;
    const Plan& p = e->plan;
    HIPCHK(hipStreamSynchronize(e->stream));
    void* old[] = {bt.d_rec[0], bt.d_rec[1], bt.d_node[0], bt.d_node[1], bt.d_frozen, bt.d_beliefs, bt.d_res_hist, bt.d_sync, bt.d_ctl, bt.d_s_state,
                   bt.d_ev, bt.d_g_state, bt.d_g_frz, bt.d_g_sync};
    if (bt.h_ev) (void)hipHostFree(bt.h_ev);
    if (bt.h_beliefs) (void)hipHostFree(bt.h_beliefs);
    for (void* q : old)
        if (q) (void)hipFree(q);
    if (bt.h_ctl) (void)hipHostFree(bt.h_ctl);
    bt = bn_engine::Batch();
    int r;
    const size_t B = size_t(n_sets);
    for (int i = 0; i < 2; ++i) {
        if ((r = dalloc(&bt.d_rec[i], B * size_t(p.rec_total_doubles)))) return r;
        if ((r = dalloc(&bt.d_node[i], B * size_t(p.node_doubles)))) return r;
        HIPCHK(hipMemsetAsync(bt.d_rec[i], 0, std::max<size_t>(B * p.rec_total_doubles, 1) * 8, e->stream));
        HIPCHK(hipMemsetAsync(bt.d_node[i], 0, std::max<size_t>(B * p.node_doubles, 1) * 8, e->stream));
    }
    if ((r = dalloc(&bt.d_frozen, B * size_t(std::max(p.n_slots, 1))))) return r;
    if ((r = dalloc(&bt.d_beliefs, B * size_t(p.node_off[p.n])))) return r;
    if ((r = dalloc(&bt.d_res_hist, B * size_t(e->res_cap)))) return r;
    if ((r = dalloc(&bt.d_sync, std::min<size_t>(B, kResidentMaxSets)))) return r;
    if ((r = dalloc(&bt.d_ctl, B))) return r;
    if (e->small_ok && (r = dalloc(&bt.d_s_state, B * size_t(2 * e->small.M + 2 * e->small.N)))) return r;
    HIPCHK(hipMemsetAsync(bt.d_ctl, 0, sizeof(Ctl) * B, e->stream));  // done_run = 0: no run is marked done
    HIPCHK(hipMemsetAsync(bt.d_frozen, 0, B * size_t(std::max(p.n_slots, 1)), e->stream));
    HIPCHK(hipMemsetAsync(bt.d_beliefs, 0, std::max<size_t>(B * p.node_off[p.n], 1) * 8, e->stream));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&bt.h_ctl), sizeof(Ctl) * B, hipHostMallocMapped));
    std::memset(bt.h_ctl, 0, sizeof(Ctl) * B);
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&bt.h_ctl_dev), bt.h_ctl, 0));
    HIPCHK(hipStreamSynchronize(e->stream));
    bt.cap_sets = n_sets;
    return BN_OK;
}

// buffers of evidence set q inside the batch arrays
static BpBuffers batch_buffers_of(bn_engine* e, int32_t q) {
    const Plan& p = e->plan;
    const bn_engine::Batch& bt = e->batch;
    BpBuffers b = buffers_of(e);
    b.rec0 = bt.d_rec[0] + size_t(q) * p.rec_total_doubles;
    b.rec1 = bt.d_rec[1] + size_t(q) * p.rec_total_doubles;
    b.node0 = bt.d_node[0] + size_t(q) * p.node_doubles;
    b.node1 = bt.d_node[1] + size_t(q) * p.node_doubles;
    b.frozen = bt.d_frozen + size_t(q) * std::max(p.n_slots, 1);
    b.frozen_mark = 1;  // batches clear their marks with a memset per call
    b.beliefs = bt.d_beliefs + size_t(q) * p.node_off[p.n];
    b.res_hist = bt.d_res_hist + size_t(q) * e->res_cap;
    b.ctl = bt.d_ctl + q;
    return b;
}

// Batches want throughput; a layout built for the latency of one query (wide lane groups, any-arity tiles for
// nodes with many children: bn_plan.cpp) has up to 4x the wavefronts.  Such an engine answers batches of two or more
// sets through a second engine built from the same model with the dense layout (lanes_per_node = 2); the networks
// this concerns are small, so the second copy is too.  Created at the first such call.
static bn_engine* dense_engine_for_batch(bn_engine* e, int32_t n_sets, int& rc) {
    rc = BN_OK;
    if (!e->plan.latency_rules_applied || e->plan.nranks > 1 || n_sets < 2) return nullptr;
    if (e->small_ok && e->small_mode != 0 && e->multisweep != 0) return nullptr;  // one workgroup per set (bn_small.hip): the layout plays no part
    if (mid_applies(e)) return nullptr;                                           // ... or a few per set (bn_mid.hip)
    if (dag_applies(e)) return nullptr;                                           // ... or the register-resident DAG path, set by set (bn_dag.hip)
    if (e->dense_refused) return nullptr;
    if (!e->dense) {
        const Plan& p = e->plan;
        bn_model_desc d;
        d.n_nodes = p.n;
        d.k = p.k.data(); d.in_ptr = p.in_ptr.data(); d.in_idx = p.in_idx.data();
        d.cpt_off = p.cpt_off.data(); d.cpt = p.cpt_flat.data();
        d.device = e->device;
        d.lanes_per_node = p.group_wide ? 4 : 2;  // same lane-group split: same bits as this engine's single queries
        rc = bn_create(&d, &e->dense);
        if (rc) { e->dense = nullptr; return nullptr; }
        // "each set gets exactly the result its single query gives it" (bn_mi355x.h): the two layouts may put a node on DIFFERENT tile
        // variants (the latency layout gives nodes with many children the any-arity tiles), and for tables beyond 128 entries -- >= 3
        // parents, or two parents of arity >= 6 -- the variants sum in different orders (each within 1e-12 of the reference, but not the
        // same bits; scripts/soak_gpu.py, round 6: a 200-node network of arities {4, 6} differed by 1e-16 between a batch and its single
        // queries).  Where that happens to some node the batch runs on this engine's own layout.
        const Plan &p0 = e->plan, &p1 = e->dense->plan;
        bool same_bits = p1.n == p0.n;
        for (int32_t v = 0; same_bits && v < p0.n; ++v) {
            if (p0.node_class[v] < 0 || p1.node_class[v] < 0) continue;
            const ClassDesc &c0 = p0.classes[p0.node_class[v]], &c1 = p1.classes[p1.node_class[v]];
            if ((c0.variant != c1.variant || c0.G != c1.G) && int64_t(c0.kv) * c0.rows > 128) same_bits = false;
        }
        if (!same_bits) {
            free_engine(e->dense);
            e->dense = nullptr;
            e->dense_refused = true;
            return nullptr;
        }
    }
    e->dense->multisweep = e->multisweep;
    e->dense->small_mode = e->small_mode;
    e->dense->mid_mode = e->mid_mode;
    e->dense->dag_mode = e->dag_mode;
    e->dense->timing = e->timing;
    return e->dense;
}
static void adopt_batch_outcome(bn_engine* e) {  // what bn_bp_stats / bn_bp_last_path report after a forwarded batch
    e->last_path = e->dense->last_path;
    const bn_bp_stats own = e->stats;
    e->stats = e->dense->stats;
    e->stats.algorithmic_bytes_per_sweep = own.algorithmic_bytes_per_sweep;
    e->stats.layout_bytes_per_sweep = own.layout_bytes_per_sweep;
    e->stats.messages_per_sweep = own.messages_per_sweep;
}

// The batch's evidence (bt.d_ev) -> the sets' tile buffers: marks cleared, one bp_evidence_kernel per set.  No-op when done already.
static int flush_batch_evidence(bn_engine* e) {
    bn_engine::Batch& bt = e->batch;
    if (!bt.ev_deferred) return BN_OK;
    const Plan& p = e->plan;
    HIPCHK(hipMemsetAsync(bt.d_frozen, 0, size_t(bt.n_sets) * size_t(std::max(p.n_slots, 1)), e->stream));
    for (int32_t q = 0; q < bt.n_sets; ++q) {
        EvidenceArgs ea{batch_buffers_of(e, q), bt.ne[q], reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_node) + bt.ev_node_at[q],
                        reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_off) + bt.ev_off_at[q],
                        reinterpret_cast<double*>(bt.ev_base + bt.ev_b_val) + bt.ev_val_at[q]};
        if (int code = launch_bp_evidence(ea, e->stream))
            return fail(BN_ERR_HIP, std::string("bp_evidence launch failed: ") + hipGetErrorString(hipError_t(code)));
    }
    bt.ev_deferred = false;
    return BN_OK;
}

extern "C" int bn_bp_set_evidence_batch(bn_engine* e, int32_t n_sets, const int32_t* ne, const int32_t* ev_node,
                                        const int32_t* ev_off, const double* ev_val) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (e->host_only) return fail(BN_ERR_STATE, "engine was created with BN_DEVICE_HOST_ONLY: no GPU, no compute");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: bn_reload_cpt failed while uploading (destroy it and create a new one)");
    e->batch_on_dense = false;
    if (n_sets >= 1 && n_sets <= BN_MAX_BATCH_SETS) {
        int rc;
        if (bn_engine* de = dense_engine_for_batch(e, n_sets, rc)) {
            rc = bn_bp_set_evidence_batch(de, n_sets, ne, ev_node, ev_off, ev_val);
            e->batch_on_dense = rc == BN_OK;
            return rc;
        } else if (rc) {
            return rc;
        }
    }
    if (n_sets < 1 || n_sets > BN_MAX_BATCH_SETS) return fail(BN_ERR_ARG, "n_sets must be in 1.." + std::to_string(BN_MAX_BATCH_SETS));
    if (e->plan.nranks > 1) return fail(BN_ERR_STATE, "batched evidence sets are not available on sharded engines");
    if (!ne) return fail(BN_ERR_ARG, "null ne");
    const Plan& p = e->plan;
    // validate every set like bn_bp_set_evidence does; locate its slices of the concatenated arrays
    std::vector<int64_t> node_at(n_sets + 1, 0), off_at(n_sets + 1, 0), val_at(n_sets + 1, 0);
    for (int32_t q = 0; q < n_sets; ++q) {
        if (ne[q] < 0) return fail(BN_ERR_ARG, "negative evidence count");
        if (ne[q] > 0 && (!ev_node || !ev_off || !ev_val)) return fail(BN_ERR_ARG, "null evidence array");
        int rc = check_evidence(p, ne[q], ev_node ? ev_node + node_at[q] : nullptr, ev_off ? ev_off + off_at[q] : nullptr, e->ev_seen, e->ev_epoch);
        if (rc) return rc;
        node_at[q + 1] = node_at[q] + ne[q];
        off_at[q + 1] = off_at[q] + ne[q] + 1;
        val_at[q + 1] = val_at[q] + (ne[q] > 0 ? ev_off[off_at[q] + ne[q]] : 0);
    }
    ON_DEVICE(e);
    int rc = batch_reserve(e, n_sets);
    if (rc) return rc;
    bn_engine::Batch& bt = e->batch;
    bt.n_sets = n_sets;
    bt.have_run = false;
    bt.ne.assign(ne, ne + n_sets);
    bt.ev_node.assign(ev_node, ev_node + node_at[n_sets]);
    bt.ev_off.assign(ev_off, ev_off + (node_at[n_sets] > 0 || ev_off ? off_at[n_sets] : 0));
    bt.ev_val.assign(ev_val, ev_val + val_at[n_sets]);
    // one staging block [nodes | offs | vals | per-set meta]
    const size_t b_node = 0, b_off = size_t(node_at[n_sets]) * 4, b_val = (b_off + size_t(off_at[n_sets]) * 4 + 7) & ~size_t(7);
    const size_t b_meta = b_val + size_t(val_at[n_sets]) * 8;
    const size_t bytes = b_meta + size_t(n_sets) * 32;
    auto fill = [&](char* dst) {
        if (node_at[n_sets] > 0) {
            std::memcpy(dst + b_node, ev_node, size_t(node_at[n_sets]) * 4);
            std::memcpy(dst + b_val, ev_val, size_t(val_at[n_sets]) * 8);
        }
        if (ev_off) std::memcpy(dst + b_off, ev_off, size_t(off_at[n_sets]) * 4);
        int32_t* meta = reinterpret_cast<int32_t*>(dst + b_meta);  // per set {count, first node entry, first offset entry, first value, values}
        for (int32_t q = 0; q < n_sets; ++q) {
            meta[8 * q] = ne[q]; meta[8 * q + 1] = int32_t(node_at[q]); meta[8 * q + 2] = int32_t(off_at[q]); meta[8 * q + 3] = int32_t(val_at[q]);
            meta[8 * q + 4] = int32_t(val_at[q + 1] - val_at[q]); meta[8 * q + 5] = meta[8 * q + 6] = meta[8 * q + 7] = 0;
        }
    };
    bt.ev_b_node = b_node; bt.ev_b_off = b_off; bt.ev_b_val = b_val;
    bt.ev_node_at = node_at; bt.ev_off_at = off_at; bt.ev_val_at = val_at;
    bt.ev_deferred = true;
    bt.dag_ev_applied = false;
    bt.beliefs_on_host = false;
    if (e->small_ok || e->mid_ok || dag_applies(e)) {
        // Small networks: the block is page-locked host memory that the kernels read in place -- the one-workgroup kernel (one
        // workgroup per set) each set's arrays, no copy command, no evidence launch per set, no synchronisation here; the tile
        // buffers get the marks and vectors only if another path runs the batch (flush_batch_evidence).  (No kernel is in
        // flight when the block is rewritten: every run entry point synchronises before it returns.)
        if (bytes > bt.h_ev_cap) {
            if (bt.h_ev) (void)hipHostFree(bt.h_ev);
            bt.h_ev = nullptr;
            bt.h_ev_cap = std::max<size_t>(bytes * 2, 4096);
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&bt.h_ev), bt.h_ev_cap, hipHostMallocMapped));
            HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&bt.ev_base), bt.h_ev, 0));
        }
        fill(bt.h_ev);
        bt.d_ev_meta = reinterpret_cast<int32_t*>(bt.ev_base + b_meta);
        return BN_OK;
    }
    // every other network: one H2D copy, then one evidence kernel per set
    if (bytes > bt.ev_cap) {
        if (bt.d_ev) (void)hipFree(bt.d_ev);
        bt.d_ev = nullptr;
        bt.ev_cap = std::max<size_t>(bytes * 2, 4096);
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&bt.d_ev), bt.ev_cap));
    }
    bt.ev_base = bt.d_ev;
    std::vector<char> host(std::max<size_t>(bytes, 1));
    fill(host.data());
    HIPCHK(hipMemcpyAsync(bt.d_ev, host.data(), bytes, hipMemcpyHostToDevice, e->stream));
    bt.d_ev_meta = reinterpret_cast<int32_t*>(bt.ev_base + b_meta);
    if ((rc = flush_batch_evidence(e))) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));  // `host` is a local
    return BN_OK;
}

// sets [first, first + count) through the resident kernel, round-robin in one launch (count <= kResidentMaxSets): enqueue only.  The
// chunks of a batch follow each other on the stream and the host waits once for all of them (a wait per chunk cost a batch of 16 sets
// four wake-ups and four launch latencies); consecutive launches share the barrier words -- the generations count on.
static int enqueue_batch_resident_chunk(bn_engine* e, double eps, int32_t max_sweeps, int32_t first, int32_t count, int32_t begin, uint32_t mask) {
    bn_engine::Batch& bt = e->batch;
    const Plan& p = e->plan;
    hipStream_t s = e->stream;
    if (bt.sync_dirty || bt.gen_base > (1u << 29)) {
        HIPCHK(hipMemsetAsync(bt.d_sync, 0, sizeof(ResidentSync) * size_t(std::min(bt.cap_sets, kResidentMaxSets)), s));
        bt.sync_dirty = false;
        bt.gen_base = 0;
    }
    ResidentArgs a{batch_buffers_of(e, first), eps, max_sweeps, begin, kResidentBudget, e->run_id, bt.gen_base, 5000000ull, bt.d_sync,
                   bt.h_ctl_dev + first, e->grid_resident, e->resident_waves, count, mask, p.rec_total_doubles, p.node_doubles,
                   int64_t(std::max(p.n_slots, 1)), p.node_off[p.n], e->res_cap, nullptr, nullptr, nullptr, 0, nullptr, 1, 0, e->h_abort_dev};
    if (int code = launch_bp_resident(a, e->grid_resident + resident_service_blocks(e->grid_resident), e->resident_lean, s))
        return fail(BN_ERR_HIP, std::string("bp_resident launch failed: ") + hipGetErrorString(hipError_t(code)));
    bt.gen_base += kResidentBudget + 1;
    return BN_OK;
}

// after the stream has drained: what the launch of sets [first, first + count) (those in `mask`) reported.  next = the sets whose run
// goes on beyond the launch's budget of iterations.
static int collect_batch_resident_chunk(bn_engine* e, int32_t first, int32_t count, uint32_t mask, uint32_t& next, double& dev_ticks) {
    bn_engine::Batch& bt = e->batch;
    next = 0;
    if (*e->h_abort != 0) {
        bt.sync_dirty = true;
        return fail(BN_ERR_STATE, "resident kernel gave up a barrier wait");
    }
    for (int32_t q = 0; q < count; ++q) {
        if (!((mask >> q) & 1u)) continue;
        const Ctl& c = bt.h_ctl[first + q];
        if (c.run_id != e->run_id || c.done < 0) bt.sync_dirty = true;
        if (c.run_id != e->run_id) return fail(BN_ERR_STATE, "resident kernel did not report (stale control block)");
        if (c.done < 0) return fail(BN_ERR_STATE, "resident kernel gave up a barrier wait");
        bt.sweeps[first + q] = c.n_sweeps;
        bt.residual[first + q] = c.last_res;
        if (c.done == 0) next |= 1u << q;
    }
    dev_ticks += double(bt.h_ctl[first].t_last - bt.h_ctl[first].t_first);
    return BN_OK;
}

// every set through the resident kernel: up to kResidentMaxSets per launch, further sets in further launches
static int run_batch_resident(bn_engine* e, double eps, int32_t max_sweeps) {
    bn_engine::Batch& bt = e->batch;
    hipStream_t s = e->stream;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    int32_t launches = 0;
    double dev_ticks = 0.0;
    float ms = 0.f;
    struct Chunk { int32_t first, count; uint32_t next; };
    std::vector<Chunk> chunks;
    const int32_t n_chunks = (bt.n_sets + kResidentMaxSets - 1) / kResidentMaxSets;
    for (int32_t c = 0, first = 0; c < n_chunks; ++c) {
        const int32_t count = (bt.n_sets - first + (n_chunks - c) - 1) / (n_chunks - c);  // balanced chunk sizes
        chunks.push_back(Chunk{first, count, 0u});
        first += count;
    }
    *e->h_abort = 0;
    for (int32_t q = 0; q < bt.n_sets; ++q) bt.h_ctl[q].run_id = 0;
    if (e->timing) {
        int rc = ensure_events(e, 2);
        if (rc) return rc;
        HIPCHK(hipEventRecord(e->events[0], s));
    }
    int rc = BN_OK;
    size_t enqueued = 0;
    for (; enqueued < chunks.size() && rc == BN_OK; ++enqueued)
        rc = enqueue_batch_resident_chunk(e, eps, max_sweeps, chunks[enqueued].first, chunks[enqueued].count, 0, (1u << chunks[enqueued].count) - 1u);
    if (rc != BN_OK) --enqueued;   // (the last one was not launched)
    if (e->timing && rc == BN_OK) HIPCHK(hipEventRecord(e->events[1], s));
    // (also after a failed enqueue: what is on the stream writes into the batch's buffers)
    const hipError_t drained = hipStreamSynchronize(s);
    if (drained != hipSuccess && rc == BN_OK) rc = fail(BN_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(drained));
    launches += int32_t(enqueued);
    if (rc != BN_OK) { bt.sync_dirty = true; *e->h_abort = 0; return rc; }
    if (e->timing) {
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, e->events[0], e->events[1]));
        ms += t;
    }
    for (Chunk& c : chunks) {
        const int rc_c = collect_batch_resident_chunk(e, c.first, c.count, (1u << c.count) - 1u, c.next, dev_ticks);
        if (rc == BN_OK) rc = rc_c;
    }
    if (*e->h_abort != 0) { *e->h_abort = 0; bt.sync_dirty = true; }
    if (rc != BN_OK) return rc;
    // runs beyond one launch's budget of iterations (rare): those sets go on, chunk by chunk, a launch and a wait at a time
    for (Chunk& c : chunks) {
        int32_t begin = 0;
        while (c.next != 0) {
            const uint32_t mask = c.next;
            begin += kResidentBudget;
            *e->h_abort = 0;
            for (int32_t q = 0; q < c.count; ++q)
                if ((mask >> q) & 1u) bt.h_ctl[c.first + q].run_id = 0;
            if ((rc = enqueue_batch_resident_chunk(e, eps, max_sweeps, c.first, c.count, begin, mask))) return rc;
            HIPCHK(hipStreamSynchronize(s));
            ++launches;
            if ((rc = collect_batch_resident_chunk(e, c.first, c.count, mask, c.next, dev_ticks))) { *e->h_abort = 0; return rc; }
        }
    }
    e->last_path = 2;
    e->stats.sweep_launches = launches;
    e->stats.sweep_kernel_ms = ms;
    e->stats.sweep_devclock_ms = float(dev_ticks * 1e-5);
    e->stats.sweeps = *std::max_element(bt.sweeps.begin(), bt.sweeps.end());
    return BN_OK;
}

// Every set in each per-sweep launch (blockIdx.y = evidence set): any tile variants.  The sets share the launch
// and its latency -- what a small or latency-bound network pays for -- and the CPT lines in the caches; each keeps
// its own records, node vectors, marks, residual slots and done mark, so it stops on the sweep its single run
// stops on (a converged set's blocks return at once in the launches the others still need).
static int run_batch_launches(bn_engine* e, double eps, int32_t max_sweeps) {
    bn_engine::Batch& bt = e->batch;
    const Plan& p = e->plan;
    hipStream_t s = e->stream;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    const int32_t B = bt.n_sets;
    if (!bt.rows_clean) {  // an earlier batched run did not end through its finish kernel
        for (int32_t q = 0; q < bt.cap_sets; ++q)
            if (int code = launch_bp_reset(batch_buffers_of(e, q), s))
                return fail(BN_ERR_HIP, std::string("bp_reset launch failed: ") + hipGetErrorString(hipError_t(code)));
    }
    bt.rows_clean = false;
    const SetStrides st{p.rec_total_doubles, p.node_doubles, int64_t(std::max(p.n_slots, 1)), p.node_off[p.n], e->res_cap};
    const BpBuffers b0 = batch_buffers_of(e, 0);
    const int32_t nt = int32_t(p.tiles.size());
    const int grid = ((nt + 1 + kWavesPerBlock - 1) / kWavesPerBlock + 7) & ~7;
    static const bool no_light = std::getenv("BN_NO_LIGHT") != nullptr;
    int32_t launched = 0;
    int32_t batch = bt.predicted_sweeps > 0 ? bt.predicted_sweeps : (e->predicted_sweeps > 0 ? e->predicted_sweeps : 8);
    for (;;) {
        if (max_sweeps > 0) batch = std::min(batch, max_sweeps - launched);
        for (int32_t i = 0; i < batch; ++i) {
            const int32_t sweep = launched + i;
            const int cur = sweep & 1;
            SweepArgs sa{b0, bt.d_rec[cur], bt.d_rec[cur ^ 1], bt.d_node[cur], bt.d_node[cur ^ 1], eps, sweep, 0, nt, 1, e->run_id, st};
            // B == 1 runs the plain instantiation on set 0's buffers
            if (launch_bp_sweep(sa, grid, B, false, p.light && !no_light, p.variants, s)) return fail(BN_ERR_HIP, "bp_sweep launch failed");
        }
        launched += batch;
        FinishArgs fa{b0, eps, launched, (max_sweeps > 0 && launched >= max_sweeps) ? 1 : 0, e->run_id, bt.h_ctl_dev, st};
        if (launch_bp_finish(fa, e->grid_tiles, B, s)) return fail(BN_ERR_HIP, "bp_finish launch failed");
        HIPCHK(hipStreamSynchronize(s));
        bool all_done = true;
        for (int32_t q = 0; q < B; ++q) {
            if (bt.h_ctl[q].run_id != e->run_id) return fail(BN_ERR_STATE, "finish kernel did not report (stale control block)");
            if (bt.h_ctl[q].done == 0) all_done = false;
        }
        if (all_done) break;
        batch = 8;
    }
    bt.rows_clean = true;  // every set's run ended in a finish kernel that saw it over
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int32_t q = 0; q < B; ++q) {
        bt.sweeps[q] = bt.h_ctl[q].n_sweeps;
        bt.residual[q] = bt.h_ctl[q].last_res;
        t0 = std::min(t0, bt.h_ctl[q].t_first);
        t1 = std::max(t1, bt.h_ctl[q].t_last);
    }
    bt.predicted_sweeps = *std::max_element(bt.sweeps.begin(), bt.sweeps.end());
    e->last_path = 0;
    e->stats.sweep_launches = launched;
    e->stats.sweep_kernel_ms = 0.f;
    e->stats.sweep_devclock_ms = t1 > t0 ? float(double(t1 - t0) * 1e-5) : 0.f;
    e->stats.sweeps = bt.predicted_sweeps;
    return BN_OK;
}

// Small networks: one workgroup per evidence set, all sets in ONE launch, each set stopping by itself (bn_small.hip).
static int run_batch_small(bn_engine* e, double eps, int32_t max_sweeps) {
    bn_engine::Batch& bt = e->batch;
    const Plan& p = e->plan;
    hipStream_t s = e->stream;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    const int32_t B = bt.n_sets;
    const SetStrides st{p.rec_total_doubles, p.node_doubles, int64_t(std::max(p.n_slots, 1)), p.node_off[p.n], e->res_cap};
    const int64_t state_stride = 2 * int64_t(e->small.M) + 2 * int64_t(e->small.N);
    SmallArgs a = small_args_of(e, batch_buffers_of(e, 0), eps, max_sweeps, 0, bt.h_ctl_dev);
    a.state = bt.d_s_state; a.sets = st; a.state_stride = state_stride;
    const size_t per_set = size_t(p.node_off[p.n]);
    if (bt.direct_out) {  // bn_bp_run_batch: the marginals go straight into page-locked host memory (no copy command, no second sync)
        if (size_t(B) * per_set > bt.h_beliefs_cap) {
            if (bt.h_beliefs) (void)hipHostFree(bt.h_beliefs);
            bt.h_beliefs = nullptr;
            bt.h_beliefs_cap = size_t(bt.cap_sets) * per_set;
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&bt.h_beliefs), std::max<size_t>(bt.h_beliefs_cap, 1) * sizeof(double), hipHostMallocMapped));
            HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&bt.h_beliefs_dev), bt.h_beliefs, 0));
        }
        a.b.beliefs = bt.h_beliefs_dev;
    }
    bt.beliefs_on_host = bt.direct_out;
    auto evidence_of = [&](SmallArgs& x, bool per_set_meta, int32_t q) {
        if (!bt.ev_deferred) return;  // the tile buffers hold it
        x.ev_mode = 1;
        x.ev_node = reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_node);
        x.ev_off = reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_off);
        x.ev_val = reinterpret_cast<double*>(bt.ev_base + bt.ev_b_val);
        x.ev_meta = per_set_meta ? bt.d_ev_meta : bt.d_ev_meta + 8 * q;  // (a single-set launch reads entry `blockIdx.x` = 0)
    };
    evidence_of(a, true, 0);
    if (int code = launch_bp_small(a, e->small.waves, e->small.lds_bytes, B, s))
        return fail(BN_ERR_HIP, std::string("bp_small launch failed: ") + hipGetErrorString(hipError_t(code)));
    HIPCHK(hipStreamSynchronize(s));
    int32_t launches = 1;
    for (int32_t q = 0; q < B; ++q) {
        if (bt.h_ctl[q].run_id != e->run_id) return fail(BN_ERR_HIP, "bp_small kernel did not report (stale control block)");
        while (bt.h_ctl[q].done == 0) {  // a set that used up the launch's budget of iterations goes on by itself
            SmallArgs c = small_args_of(e, batch_buffers_of(e, q), eps, max_sweeps, bt.h_ctl[q].n_sweeps, bt.h_ctl_dev + q);
            c.state = bt.d_s_state + size_t(q) * state_stride;
            if (bt.direct_out) c.b.beliefs = bt.h_beliefs_dev + size_t(q) * per_set;
            evidence_of(c, false, q);
            if (int code = launch_bp_small(c, e->small.waves, e->small.lds_bytes, 1, s))
                return fail(BN_ERR_HIP, std::string("bp_small launch failed: ") + hipGetErrorString(hipError_t(code)));
            HIPCHK(hipStreamSynchronize(s));
            ++launches;
        }
    }
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int32_t q = 0; q < B; ++q) {
        bt.sweeps[q] = bt.h_ctl[q].n_sweeps;
        bt.residual[q] = bt.h_ctl[q].last_res;
        t0 = std::min(t0, bt.h_ctl[q].t_first);
        t1 = std::max(t1, bt.h_ctl[q].t_last);
    }
    bt.predicted_sweeps = *std::max_element(bt.sweeps.begin(), bt.sweeps.end());
    e->last_path = 3;
    e->stats.sweep_launches = launches;
    e->stats.sweep_kernel_ms = 0.f;
    e->stats.sweep_devclock_ms = t1 > t0 ? float(double(t1 - t0) * 1e-5) : 0.f;
    e->stats.sweeps = bt.predicted_sweeps;
    return BN_OK;
}

// Mid-size networks: every set runs exactly like a single query (same kernel, same bits), as many sets per launch as fit the
// chip with a workgroup per CU (the grid barrier needs every workgroup of a set resident).  BN_ERR_STATE: a grid wait gave up.
static int run_batch_mid(bn_engine* e, double eps, int32_t max_sweeps) {
    bn_engine::Batch& bt = e->batch;
    const Plan& p = e->plan;
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    const int32_t B = bt.n_sets, nparts = int32_t(e->mid.parts.size());
    const int32_t per_launch = std::max(1, std::min(B, (e->n_cus * 9 / 10) / nparts));
    int rc;
    if ((rc = mid_reserve_slots(e, per_launch))) return rc;
    const SetStrides st{p.rec_total_doubles, p.node_doubles, int64_t(std::max(p.n_slots, 1)), p.node_off[p.n], e->res_cap};
    const BpBuffers b0 = batch_buffers_of(e, 0);
    auto evidence_of = [&](MidArgs& x) {
        if (!bt.ev_deferred) return;  // the tile buffers hold it
        x.ev_mode = 1;
        x.ev_node = reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_node);
        x.ev_off = reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_off);
        x.ev_val = reinterpret_cast<double*>(bt.ev_base + bt.ev_b_val);
        x.ev_meta = bt.d_ev_meta;
    };
    int32_t launches = 0;
    // the chunks follow each other on the stream (a chunk's sets use the state slots the previous chunk's kernel has left), one wait
    *e->h_abort = 0;
    for (int32_t first = 0; first < B && rc == BN_OK; first += per_launch) {
        MidArgs a = mid_args_of(e, b0, st, bt.h_ctl_dev, eps, max_sweeps, 0, first, 0);
        evidence_of(a);
        rc = mid_launch(e, a, std::min(per_launch, B - first), nullptr, nullptr, false);
        if (rc == BN_OK) ++launches;
    }
    {   // (also after a failed enqueue: what is on the stream writes into the batch's buffers)
        const hipError_t drained = hipStreamSynchronize(e->stream);
        if (drained != hipSuccess && rc == BN_OK) rc = fail(BN_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(drained));
    }
    e->ev_upload_pending = false;
    if (*e->h_abort != 0) {
        *e->h_abort = 0;
        if (rc == BN_OK) rc = fail(BN_ERR_STATE, "a workgroup of the mid-size kernel gave up its grid wait");
    }
    if (rc != BN_OK) return rc;
    for (int32_t first = 0; first < B; first += per_launch) {
        const int32_t count = std::min(per_launch, B - first);
        for (int32_t q = first; q < first + count; ++q) {
            if (bt.h_ctl[q].done < 0) return fail(BN_ERR_STATE, "a workgroup of the mid-size kernel gave up its grid wait");
            if (bt.h_ctl[q].run_id != e->run_id) return fail(BN_ERR_HIP, "bp_mid kernel did not report (stale control block)");
            while (bt.h_ctl[q].done == 0) {  // a set that used up the launch's budget of iterations goes on by itself, in its slot
                MidArgs c = mid_args_of(e, b0, st, bt.h_ctl_dev, eps, max_sweeps, bt.h_ctl[q].n_sweeps, q, q - first);
                evidence_of(c);
                if ((rc = mid_launch(e, c, 1, nullptr, nullptr))) return rc;
                ++launches;
                if (bt.h_ctl[q].done < 0) return fail(BN_ERR_STATE, "a workgroup of the mid-size kernel gave up its grid wait");
            }
        }
    }
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int32_t q = 0; q < B; ++q) {
        bt.sweeps[q] = bt.h_ctl[q].n_sweeps;
        bt.residual[q] = bt.h_ctl[q].last_res;
        t0 = std::min(t0, bt.h_ctl[q].t_first);
        t1 = std::max(t1, bt.h_ctl[q].t_last);
    }
    bt.predicted_sweeps = *std::max_element(bt.sweeps.begin(), bt.sweeps.end());
    e->last_path = 4;
    e->stats.sweep_launches = launches;
    e->stats.sweep_kernel_ms = 0.f;
    e->stats.sweep_devclock_ms = t1 > t0 ? float(double(t1 - t0) * 1e-5) : 0.f;
    e->stats.sweeps = bt.predicted_sweeps;
    return BN_OK;
}

// The register-resident DAG path (bn_dag.hip) answers a batch one set after another: every set is a single query's launch -- the
// same kernel, the same bits -- reading its evidence from the batch's staging block and writing its marginals and residual history
// into the set's slots.  BN_ERR_STATE: a grid wait gave up.
// Sets [first, first + count) of the batch in ONE launch of the register-resident DAG kernel: the sets take turns inside an
// iteration, so a set's barrier completes while the others sweep, and one set of CPT registers serves them all (bn_dag.hip,
// dag_drive).  Every set has its own state, marks, barrier words, residual history and control block and keeps the bits and the
// sweep count of its single run.  left[q] = true: set q did not finish here (more than kDagBudget sweeps) and is run on its own.
// BN_ERR_STATE: a grid wait gave up.
// (enqueue only: the chunks of a batch follow each other on the stream -- the next chunk's evidence lands in the state slots when the
// previous chunk's kernel has left them -- and the host waits once, for all of them: collect_batch_dag_chunk reads the outcome.)
struct DagChunk { int32_t first, count; uint32_t run_id; };
static int enqueue_batch_dag_chunk(bn_engine* e, double eps, int32_t max_sweeps, int32_t first, int32_t count, DagChunk& chunk) {
    bn_engine::Batch& bt = e->batch;
    const Plan& p = e->plan;
    const DagPlan& dp = e->dag;
    hipStream_t s = e->stream;
    const size_t state_d = size_t(dag_state_doubles(dp.E, dp.n));
    if (bt.dag_sets < kDagMaxSets) {   // first use: every set's state, marks and barrier words
        int r;
        if ((r = dalloc(&bt.d_g_state, state_d * kDagMaxSets))) return r;
        if ((r = dalloc(&bt.d_g_frz, size_t(dp.n) * kDagMaxSets))) return r;
        if ((r = dalloc(&bt.d_g_sync, size_t(kDagMaxSets)))) return r;
        HIPCHK(hipMemsetAsync(bt.d_g_state, 0, state_d * kDagMaxSets * sizeof(double), s));
        HIPCHK(hipMemsetAsync(bt.d_g_frz, 0, size_t(dp.n) * kDagMaxSets, s));
        bt.dag_sets = kDagMaxSets;
        bt.dag_mark = 0;
        bt.dag_sync_dirty = true;
    }
    // The evidence of a batch that fits the state slots (one chunk) on a network without padding stays where the first run put it: the
    // sweeps carry an observed node's vectors over and sweep 0 reads nothing else of the old state (the single query's dag_ev_applied).
    const bool keeps = dp.uniform4 && first == 0 && count == bt.n_sets;
    const bool apply = !(keeps && bt.dag_ev_applied);
    bt.dag_ev_applied = false;   // (true again only once every launch of this chunk is on the stream: an error return below leaves no claim behind)
    if (apply) {
        if (bt.dag_mark == 255) {  // the mark values are used up: start over
            HIPCHK(hipMemsetAsync(bt.d_g_frz, 0, size_t(dp.n) * kDagMaxSets, s));
            bt.dag_mark = 0;
        }
        ++bt.dag_mark;
    }
    if (apply) {   // pi(v) = lambda(v) = the given vector in both buffers, node marked (:68-73): every set of the chunk in one launch
        DagEvidenceBatch eb{};
        DagInitBatch ib{};
        for (int32_t q = 0; q < count; ++q) {
            const int32_t g = first + q;
            eb.set[q] = DagEvidenceArgs{bt.ne[g], dp.n, dp.E, reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_node) + bt.ev_node_at[g],
                                        reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_off) + bt.ev_off_at[g],
                                        reinterpret_cast<double*>(bt.ev_base + bt.ev_b_val) + bt.ev_val_at[g], bt.d_g_state + size_t(q) * state_d,
                                        bt.d_g_frz + size_t(q) * dp.n, bt.dag_mark, e->d_g_k, e->d_g_nperm};
            ib.set[q] = DagInitArgs{dp.n, dp.E, e->d_g_inptr, e->d_g_inidx, e->d_g_k, e->d_g_init, bt.d_g_state + size_t(q) * state_d,
                                    bt.d_g_frz + size_t(q) * dp.n, bt.dag_mark, e->d_g_eperm, e->d_g_nperm};
        }
        if (int code = launch_dag_evidence_batch(eb, count, s))
            return fail(BN_ERR_HIP, std::string("dag_evidence launch failed: ") + hipGetErrorString(hipError_t(code)));
        if (!dp.uniform4) {
            if (int code = launch_dag_init_batch(ib, count, s))
                return fail(BN_ERR_HIP, std::string("dag_init launch failed: ") + hipGetErrorString(hipError_t(code)));
        }
    }
    if (bt.dag_sync_dirty || bt.dag_gen_base > (1u << 29)) {
        HIPCHK(hipMemsetAsync(bt.d_g_sync, 0, sizeof(ResidentSync) * size_t(kDagMaxSets), s));
        bt.dag_sync_dirty = false;
        bt.dag_gen_base = 0;
    }
    ++e->run_id;
    if (e->run_id == 0) e->run_id = 1;
    chunk = DagChunk{first, count, e->run_id};
    DagArgs a{};
    a.b = buffers_of(e);
    a.b.beliefs = bt.d_beliefs + size_t(first) * p.node_off[p.n];
    a.b.res_hist = bt.d_res_hist + size_t(first) * e->res_cap;
    a.eps = eps; a.max_sweeps = max_sweeps; a.sweep_begin = 0; a.budget = kDagBudget; a.run_id = e->run_id;
    a.gen_base = bt.dag_gen_base;
    a.timeout_ticks = 5000000ull;
    a.sync = bt.d_g_sync; a.host_ctl = bt.h_ctl_dev + first; a.host_abort = e->h_abort_dev;
    a.n = dp.n; a.E = dp.E; a.n_blocks = dp.blocks;
    a.tiles = e->d_g_tiles; a.slot_ptr = e->d_g_slotptr; a.cnode = e->d_g_cnode; a.pitem = e->d_g_pitem; a.oedge = e->d_g_oedge;
    a.cpt_img = e->d_g_cpt; a.npi_init = e->d_g_init; a.state = bt.d_g_state; a.frz = bt.d_g_frz; a.frz_mark = bt.dag_mark;
    static const int poll_sleep = std::getenv("BN_DAG_SLEEP") ? std::atoi(std::getenv("BN_DAG_SLEEP")) : 1;
    static const int first_delay = std::getenv("BN_DAG_DELAY") ? std::atoi(std::getenv("BN_DAG_DELAY")) : 30;
    a.poll_sleep = poll_sleep;
    a.first_poll_delay = first_delay;
    a.n_sets = count; a.set_mask = (1u << count) - 1u;
    a.state_init = dp.uniform4 ? 0 : 1; a.node_k = e->d_g_k; a.node_off = e->d_g_noff;
    a.state_stride = int64_t(state_d); a.frz_stride = dp.n; a.belief_stride = p.node_off[p.n]; a.res_hist_stride = e->res_cap;
    for (int32_t q = 0; q < count; ++q) bt.h_ctl[first + q].run_id = 0;
    if (int code = launch_bp_dag(a, dp.stream, s))
        return fail(BN_ERR_HIP, std::string("bp_dag launch failed: ") + hipGetErrorString(hipError_t(code)));
    bt.dag_gen_base += kDagBudget + 1;
    bt.dag_ev_applied = keeps;
    return BN_OK;
}

// after the stream has drained.  left[q] = true: set q did not finish in its launch (more than kDagBudget sweeps) and is run on its own.
static int collect_batch_dag_chunk(bn_engine* e, const DagChunk& chunk, std::vector<char>& left, double& dev_ms, int32_t& max_sw) {
    bn_engine::Batch& bt = e->batch;
    const int32_t first = chunk.first, count = chunk.count;
    bool gave_up = *e->h_abort != 0, stale = false;
    for (int32_t q = 0; q < count; ++q) {
        gave_up = gave_up || bt.h_ctl[first + q].done < 0;
        stale = stale || bt.h_ctl[first + q].run_id != chunk.run_id;
    }
    if (gave_up || stale) { bt.dag_sync_dirty = true; bt.dag_ev_applied = false; }
    if (gave_up) return fail(BN_ERR_STATE, "a block of the register-resident DAG kernel gave up its grid wait");
    if (stale) return fail(BN_ERR_HIP, "bp_dag kernel did not report (stale control block)");
    dev_ms += double(bt.h_ctl[first].t_last - bt.h_ctl[first].t_first) * 1e-5;
    for (int32_t q = 0; q < count; ++q) {
        const Ctl& c = bt.h_ctl[first + q];
        if (c.done == 0) { left[first + q] = 1; continue; }   // the budget of one launch ran out: this set goes on alone
        bt.sweeps[first + q] = c.n_sweeps;
        bt.residual[first + q] = c.last_res;
        max_sw = std::max(max_sw, c.n_sweeps);
    }
    return BN_OK;
}

static int run_batch_dag(bn_engine* e, double eps, int32_t max_sweeps) {
    if (int rc0 = ensure_dag(e)) return rc0;   // (first use of the path on this engine)
    bn_engine::Batch& bt = e->batch;
    const Plan& p = e->plan;
    int rc = BN_OK;
    int32_t launches = 0, max_sw = 0;
    double dev_ms = 0.0;
    std::vector<char> left(size_t(bt.n_sets), 0);
    // how many sets share a launch (BN_DAG_SETS, default 16; 1 = one after another).  Config 2, us per set-sweep at B = 16: 8.7 / 6.9 / 6.2 / 5.9
    // with 1 / 2 / 4 / 8 sets per launch in round 4; round 5 (a turn's arrival behind the next turn's loads, one evidence launch per
    // chunk, one host wait): 4.65 with 8, 4.49 with 16 -- the pace inside the kernel is a wave's set-turn, but a launch's ramp, its
    // evidence launch and the tail where few sets are left come once instead of twice (scripts/time_dag_batch.py)
    static const int per_launch = std::max(1, std::min(kDagMaxSets, std::getenv("BN_DAG_SETS") ? std::atoi(std::getenv("BN_DAG_SETS")) : kDagMaxSets));
    if (per_launch > 1 && bt.n_sets > 1) {
        std::vector<DagChunk> chunks;
        *e->h_abort = 0;
        for (int32_t first = 0; first < bt.n_sets && rc == BN_OK; first += per_launch) {
            chunks.emplace_back();
            rc = enqueue_batch_dag_chunk(e, eps, max_sweeps, first, std::min(per_launch, bt.n_sets - first), chunks.back());
            if (rc != BN_OK) chunks.pop_back();
        }
        // (also after a failed enqueue: what is on the stream writes into the batch's buffers)
        const hipError_t drained = hipStreamSynchronize(e->stream);
        if (drained != hipSuccess && rc == BN_OK) rc = fail(BN_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(drained));
        if (rc != BN_OK) bt.dag_ev_applied = false;   // (a failed enqueue or drain: the state slots may not hold this batch's evidence)
        launches += int32_t(chunks.size());
        for (const DagChunk& c : chunks) {
            const int rc_c = collect_batch_dag_chunk(e, c, left, dev_ms, max_sw);
            if (rc == BN_OK) rc = rc_c;
        }
        if (*e->h_abort != 0) { *e->h_abort = 0; bt.dag_sync_dirty = true; }
        if (rc != BN_OK) return rc;
    } else {
        std::fill(left.begin(), left.end(), 1);
    }
    // sets left over (a run beyond one launch's budget; a batch of one): through the single-query path, one after another
    const int32_t keep_ne = e->ev_ne;
    int32_t* const keep_node = e->d_ev_node;
    int32_t* const keep_off = e->d_ev_off;
    double* const keep_val = e->d_ev_val;
    double* const keep_override = e->beliefs_override;
    bool any_left = false;
    for (int32_t q = 0; q < bt.n_sets && rc == BN_OK; ++q) {
        if (!left[q]) continue;
        any_left = true;
        e->ev_ne = bt.ne[q];
        e->d_ev_node = reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_node) + bt.ev_node_at[q];
        e->d_ev_off = reinterpret_cast<int32_t*>(bt.ev_base + bt.ev_b_off) + bt.ev_off_at[q];
        e->d_ev_val = reinterpret_cast<double*>(bt.ev_base + bt.ev_b_val) + bt.ev_val_at[q];
        e->dag_ev_applied = false;
        e->beliefs_override = bt.d_beliefs + size_t(q) * p.node_off[p.n];
        rc = run_dag(e, eps, max_sweeps, nullptr);
        if (rc != BN_OK) break;
        bt.sweeps[q] = e->last_ctl.n_sweeps;
        bt.residual[q] = e->last_ctl.last_res;
        const int32_t cnt = std::min(e->last_ctl.n_sweeps, e->res_cap);
        if (cnt > 0)
            HIPCHK(hipMemcpyAsync(bt.d_res_hist + size_t(q) * e->res_cap, e->d_res_hist, sizeof(double) * cnt, hipMemcpyDeviceToDevice, e->stream));
        launches += e->stats.sweep_launches;
        dev_ms += e->stats.sweep_devclock_ms;
        max_sw = std::max(max_sw, e->last_ctl.n_sweeps);
    }
    if (any_left) {
        // the single-query evidence in force is what the engine's own staging block holds: applied again at its next run
        e->ev_ne = keep_ne; e->d_ev_node = keep_node; e->d_ev_off = keep_off; e->d_ev_val = keep_val;
        e->dag_ev_applied = false;
        e->beliefs_override = keep_override;
    }
    if (rc != BN_OK) return rc;
    if (any_left) HIPCHK(hipStreamSynchronize(e->stream));   // (the copies of the left-over sets' residual histories; the chunks were waited for above)
    bt.predicted_sweeps = max_sw;
    e->last_path = 5;
    e->stats.sweep_launches = launches;
    e->stats.sweep_kernel_ms = 0.f;
    e->stats.sweep_devclock_ms = float(dev_ms);
    e->stats.sweeps = max_sw;
    return BN_OK;
}

// ---- the one-launch paths of a batch (bn_bp_run_batch_device): the PathDriver table of single queries, batch forms ----------------
// Which way a batch goes (measured, scripts/time_batch.py, us per set-sweep at the best batch size of either path): a small network runs one
// workgroup per set; otherwise the register-resident DAG path and the several-workgroup item kernel where their single-query policy
// chooses them; the resident tiles from ~900 tiles up -- per-sweep launches with one set per blockIdx.y share the launch latency among
// the sets, which is what smaller networks pay for (128x128 grid: 1.8 vs 7.7 resident, 200x200: 5.0 vs 8.0); on larger ones the CPT
// traffic the resident kernel saves weighs more (250x250: 9.1 vs 8.2, 316x316: 14.6 vs 8.6).  "multisweep" 2 forces the resident
// kernel wherever eligible, 0 the launches; "dag" 2 puts the DAG path in front of the one-workgroup path, as for single queries.
static bool batch_small_wanted(const bn_engine* e) {
    return e->small_ok && e->small_mode != 0 && e->multisweep != 0 && e->batch.d_s_state != nullptr && !(e->dag_mode == 2 && e->dag_ok);
}
static bool batch_dag_wanted(const bn_engine* e) { return !batch_small_wanted(e) && dag_applies(e) && e->batch.ev_base != nullptr && e->plan.nranks == 1; }
static bool batch_mid_wanted(const bn_engine* e) { return !batch_small_wanted(e) && mid_applies(e) && e->batch.ev_base != nullptr; }
static bool batch_resident_wanted(const bn_engine* e) {
    constexpr int64_t kResidentBatchMinTiles = 900;
    if (batch_small_wanted(e)) return false;
    return e->resident_ok && (e->multisweep == 2 || (e->multisweep == 1 && int64_t(e->plan.tiles.size()) >= kResidentBatchMinTiles));
}
static int run_batch_small_d(bn_engine* e, double eps, int32_t max_sweeps, double*) { return run_batch_small(e, eps, max_sweeps); }
static int run_batch_dag_d(bn_engine* e, double eps, int32_t max_sweeps, double*) { return run_batch_dag(e, eps, max_sweeps); }
static int run_batch_mid_d(bn_engine* e, double eps, int32_t max_sweeps, double*) { return run_batch_mid(e, eps, max_sweeps); }
static int run_batch_resident_d(bn_engine* e, double eps, int32_t max_sweeps, double*) { return run_batch_resident(e, eps, max_sweeps); }
static int batch_dag_gave_up(bn_engine* e) {
    ++e->dag_aborts;
    e->dag_cooldown = 64;
    report_abort_once(e, "the register-resident DAG kernel (bn_dag.hip, batch)", 64);
    return BN_OK;
}
static int batch_mid_gave_up(bn_engine* e) {
    ++e->mid_aborts;
    e->mid_cooldown = 64;
    report_abort_once(e, "the several-workgroup item kernel (bn_mid.hip, batch)", 64);
    return BN_OK;
}
static int batch_resident_gave_up(bn_engine* e) {
    ++e->resident_aborts;
    e->resident_cooldown = e->resident_backoff;
    e->resident_backoff = std::min(e->resident_backoff * 2, 1024);
    report_abort_once(e, "the resident-tile kernel (bn_resident.hip, batch)", e->resident_cooldown);
    return BN_OK;
}
static const PathDriver kBatchPaths[] = {
    {3, batch_small_wanted, run_batch_small_d, small_gave_up, nullptr, &bn_engine::small_cooldown, false},
    {5, batch_dag_wanted, run_batch_dag_d, batch_dag_gave_up, nullptr, &bn_engine::dag_cooldown, false},
    {4, batch_mid_wanted, run_batch_mid_d, batch_mid_gave_up, nullptr, &bn_engine::mid_cooldown, false},
    {2, batch_resident_wanted, run_batch_resident_d, batch_resident_gave_up, resident_ran_ok, &bn_engine::resident_cooldown, true},
};

extern "C" int bn_bp_run_batch_device(bn_engine* e, double eps, int32_t max_sweeps, int32_t* sweeps_out, double* residual_out) {
    if (!e) return fail(BN_ERR_ARG, "null engine");
    if (e->batch_on_dense && e->dense) {
        e->dense->multisweep = e->multisweep;
        e->dense->small_mode = e->small_mode;
        e->dense->mid_mode = e->mid_mode;
        e->dense->dag_mode = e->dag_mode;
        const int rc = bn_bp_run_batch_device(e->dense, eps, max_sweeps, sweeps_out, residual_out);
        if (rc == BN_OK) adopt_batch_outcome(e);
        return rc;
    }
    if (e->host_only) return fail(BN_ERR_STATE, "engine was created with BN_DEVICE_HOST_ONLY: no GPU, no compute");
    if (e->poisoned) return fail(BN_ERR_STATE, "engine unusable: bn_reload_cpt failed while uploading (destroy it and create a new one)");
    if (max_sweeps < 0) return fail(BN_ERR_ARG, "max_sweeps < 0");
    bn_engine::Batch& bt = e->batch;
    if (bt.n_sets < 1) return fail(BN_ERR_STATE, "call bn_bp_set_evidence_batch first");
    const auto t_begin = std::chrono::steady_clock::now();
    ON_DEVICE(e);
    bt.sweeps.assign(bt.n_sets, 0);
    bt.residual.assign(bt.n_sets, 0.0);
    bt.beliefs_on_host = false;
    // The one-launch paths of a batch, in the order of kBatchPaths (the same drivers' table as a single query's, with the batch forms of
    // wanted / run): the first that wants the batch and is not paused runs every set; one that gives up a bounded wait pauses itself,
    // and the whole batch is run again by the next; what none of them takes runs with one launch per sweep, one set per blockIdx.y.
    int rc = BN_ERR_STATE;
    bool evidence_flushed = false, restage = false;
    for (const PathDriver& d : kBatchPaths) {
        if (!d.wanted(e)) continue;
        if (d.reads_tile_evidence && !evidence_flushed) {   // the tile kernels read the sets' evidence from their own buffers
            if ((rc = flush_batch_evidence(e))) return rc;
            evidence_flushed = true;
        }
        int32_t& cooldown = e->*(d.cooldown);
        if (cooldown > 0) { --cooldown; rc = BN_ERR_STATE; continue; }   // paused after a launch that gave up
        rc = d.run(e, eps, max_sweeps, nullptr);
        if (rc == BN_OK) {
            if (d.ran_ok) d.ran_ok(e);
            break;
        }
        if (rc != BN_ERR_STATE) return rc;
        if (int g = d.gave_up(e)) return g;   // counters, pause, one line on stderr
        bt.sweeps.assign(bt.n_sets, 0);       // the whole batch again on the next path
        bt.residual.assign(bt.n_sets, 0.0);
        restage = restage || d.reads_tile_evidence;
    }
    if (rc != BN_OK) {
        if (restage) {
            // marks / vectors possibly half-written by an aborted launch of the tile kernels: apply every set's evidence again
            std::vector<int32_t> ne = bt.ne, ev_node = bt.ev_node, ev_off = bt.ev_off;
            std::vector<double> ev_val = bt.ev_val;
            rc = bn_bp_set_evidence_batch(e, int32_t(ne.size()), ne.data(), ev_node.data(), ev_off.data(), ev_val.data());
            if (rc) return rc;
            bt.sweeps.assign(bt.n_sets, 0);
            bt.residual.assign(bt.n_sets, 0.0);
        }
        if ((rc = flush_batch_evidence(e))) return rc;
        rc = run_batch_launches(e, eps, max_sweeps);
        if (rc) return rc;
    }
    bt.have_run = true;
    e->stats.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    for (int32_t q = 0; q < bt.n_sets; ++q) {
        if (sweeps_out) sweeps_out[q] = bt.sweeps[q];
        if (residual_out) residual_out[q] = bt.residual[q];
    }
    return BN_OK;
}

extern "C" int bn_bp_copy_beliefs_batch(bn_engine* e, double* beliefs_out) {
    if (!e || !beliefs_out) return fail(BN_ERR_ARG, "null argument");
    if (e->batch_on_dense && e->dense) return bn_bp_copy_beliefs_batch(e->dense, beliefs_out);
    if (e->host_only || !e->batch.have_run) return fail(BN_ERR_STATE, "no batched run to copy from");
    if (e->batch.beliefs_on_host) {  // the last run wrote them into page-locked host memory
        std::memcpy(beliefs_out, e->batch.h_beliefs, sizeof(double) * size_t(e->batch.n_sets) * e->plan.node_off[e->plan.n]);
        return BN_OK;
    }
    ON_DEVICE(e);
    HIPCHK(hipMemcpyAsync(beliefs_out, e->batch.d_beliefs, sizeof(double) * size_t(e->batch.n_sets) * e->plan.node_off[e->plan.n],
                          hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return BN_OK;
}

extern "C" int bn_bp_residual_history_batch(bn_engine* e, int32_t set, double* out, int32_t cap) {
    if (!e || !out || cap < 0) return fail(BN_ERR_ARG, "bad argument");
    if (e->batch_on_dense && e->dense) return bn_bp_residual_history_batch(e->dense, set, out, cap);
    if (e->host_only || !e->batch.have_run) return fail(BN_ERR_STATE, "no batched run yet");
    if (set < 0 || set >= e->batch.n_sets) return fail(BN_ERR_ARG, "set index out of range");
    const int32_t cnt = std::min({cap, e->batch.sweeps[set], e->res_cap});
    ON_DEVICE(e);
    if (cnt > 0)
        HIPCHK(hipMemcpy(out, e->batch.d_res_hist + size_t(set) * e->res_cap, sizeof(double) * cnt, hipMemcpyDeviceToHost));
    return cnt;
}

extern "C" int bn_bp_run_batch(bn_engine* e, int32_t n_sets, const int32_t* ne, const int32_t* ev_node, const int32_t* ev_off,
                               const double* ev_val, double eps, int32_t max_sweeps, double* beliefs_out, int32_t* sweeps_out,
                               double* residual_out) {
    if (!beliefs_out) return fail(BN_ERR_ARG, "null beliefs_out");
    int rc = bn_bp_set_evidence_batch(e, n_sets, ne, ev_node, ev_off, ev_val);
    if (rc) return rc;
    bn_engine* on = (e->batch_on_dense && e->dense) ? e->dense : e;
    on->batch.direct_out = true;   // (the one-workgroup path writes the marginals into page-locked host memory; other paths ignore it)
    rc = bn_bp_run_batch_device(e, eps, max_sweeps, sweeps_out, residual_out);
    on->batch.direct_out = false;
    if (rc) return rc;
    return bn_bp_copy_beliefs_batch(e, beliefs_out);
}


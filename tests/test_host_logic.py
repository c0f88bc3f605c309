"""CPU-only tests: the C ABI library loads and exports every symbol include/bn_mi355x.h declares,
argument validation, and the host-side layout / partition plan (no compute: there is no GPU here
and the product has no CPU path)."""
import os
import re

import numpy as np
import pytest

from bayesiannetwork_amd import Evidence, from_parent_lists, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def E(bnlib):
    from bayesiannetwork_amd import _lib, engine
    return lambda m, **kw: engine.Engine(m, device=_lib.BN_DEVICE_HOST_ONLY, **kw)


def test_header_symbols_are_exported(bnlib):
    from bayesiannetwork_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "bn_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, f"header vs python binding mismatch: {declared ^ bound}"
    for name in declared:
        assert hasattr(bnlib, name), f"libbn_mi355x.so does not export {name}"
    assert b"gfx950" in bnlib.bn_version()


def test_no_compute_without_gpu(E, bnlib):
    from bayesiannetwork_amd import _lib
    e = E(synth.pearl())
    with pytest.raises(_lib.BnError) as ei:
        e.bp_run(None, 1e-3)
    assert ei.value.code == _lib.BN_ERR_STATE
    with pytest.raises(_lib.BnError):
        e.lw_run(np.full(4, -1, np.int32), 10, 1)


def test_model_validation_errors(bnlib):
    from bayesiannetwork_amd import FlatModel, _lib, engine
    m = synth.pearl()

    def bad(**kw):
        d = dict(k=m.k.copy(), in_ptr=m.in_ptr.copy(), in_idx=m.in_idx.copy(), cpt_off=m.cpt_off.copy(), cpt=m.cpt.copy())
        d.update(kw)
        with pytest.raises(_lib.BnError) as ei:
            engine.Engine(FlatModel(**d), device=_lib.BN_DEVICE_HOST_ONLY)
        assert ei.value.code == _lib.BN_ERR_ARG
    bad(k=np.array([2, 2, 0, 2], np.int32))                        # arity 0
    bad(in_idx=np.array([0, 1, 0], np.int32))                      # parents of H not ascending
    bad(in_idx=np.array([0, 0, 7], np.int32))                      # parent out of range
    bad(in_idx=np.array([2, 0, 1], np.int32))                      # node 2 its own parent
    bad(cpt_off=np.array([0, 2, 4, 8, 15], np.int64))              # the reference's missing-row UB -> error
    # the stated limits of the input domain (the reference's graph_t / cpt_t have none, graph.hpp:57-154): at most
    # BN_MAX_PARENTS = 16 parents per node, arities 1..255 -- refused with a message that names the node and the limit
    def limit(model_kw, text):
        with pytest.raises(_lib.BnError, match=text) as ei:
            engine.Engine(FlatModel(**model_kw), device=_lib.BN_DEVICE_HOST_ONLY)
        assert ei.value.code == _lib.BN_ERR_ARG
    n = 18                                                          # 17 binary roots and a node with all of them as parents
    k = np.full(n, 2, np.int32)
    in_ptr = np.zeros(n + 1, np.int32)
    in_ptr[n] = 17
    cpt_off = np.concatenate([np.arange(0, 2 * 17 + 1, 2), [2 * 17 + 2 ** 18]]).astype(np.int64)
    limit(dict(k=k, in_ptr=in_ptr, in_idx=np.arange(17, dtype=np.int32), cpt_off=cpt_off, cpt=np.full(int(cpt_off[-1]), 0.5)),
          r"node 17 has 17 parents \(max 16\)")
    limit(dict(k=np.array([256], np.int32), in_ptr=np.zeros(2, np.int32), in_idx=np.zeros(0, np.int32),
               cpt_off=np.array([0, 256], np.int64), cpt=np.full(256, 1 / 256)), r"selectable_num of node 0 outside \[1,255\]")
    ok = FlatModel(k=np.array([255], np.int32), in_ptr=np.zeros(2, np.int32), in_idx=np.zeros(0, np.int32),
                   cpt_off=np.array([0, 255], np.int64), cpt=np.full(255, 1 / 255))
    engine.Engine(ok, device=_lib.BN_DEVICE_HOST_ONLY).close()     # arity 255 is inside
    with pytest.raises(_lib.BnError):
        engine.Engine(m, device=_lib.BN_DEVICE_HOST_ONLY, rank=3, nranks=2)
    with pytest.raises(_lib.BnError):
        engine.Engine(m, device=_lib.BN_DEVICE_HOST_ONLY, lanes_per_node=7)  # layout selector outside 0..4


def test_layout_invariants(E):
    for model in (synth.grid(23, 31, 4, seed=1), synth.random_dag(700, 4, 32, [2, 3, 4], seed=2), synth.pearl()):
        e = E(model)
        li = e.layout()
        assert li["n_nodes"] == model.n and li["n_edges"] == model.n_edges
        assert li["algorithmic_bytes_per_sweep"] == model.algorithmic_bytes_per_sweep()
        assert li["messages_per_sweep"] == model.messages_per_sweep()
        assert li["layout_bytes_per_sweep"] >= li["algorithmic_bytes_per_sweep"]
        slots = e.node_slots()
        assert len(set(slots.tolist())) == model.n and slots.min() >= 0   # integer node indexing is a bijection
        cls = e.layout_classes()
        assert sum(c["n_nodes"] for c in cls) == model.n
        pi, lam = e.edge_refs()
        assert (pi >= 0).all() and (lam > pi).all()                       # unsharded: every record is tile-resident
        assert len(set(pi.tolist())) == model.n_edges


def test_config3_bytes_match_survey(E):
    """316x316 grid -> 89.15 MB per sweep, 398 160 messages, 223.9 B per message."""
    g = synth.grid(316, 316, 4, seed=2)
    li = E(g).layout()
    assert li["messages_per_sweep"] == 398160
    assert abs(li["algorithmic_bytes_per_sweep"] / 1e6 - 89.15) < 0.01
    assert abs(li["algorithmic_bytes_per_sweep"] / li["messages_per_sweep"] - 223.9) < 0.05


@pytest.mark.parametrize("nranks", [2, 5, 8])
def test_partition_plan_consistent_across_ranks(E, nranks):
    """Every rank derives the exchange layout independently; they must agree, and every message
    half of a cut edge must have exactly one slot that no other half shares."""
    model = synth.random_dag(1500, 4, 64, [4, 3, 2, 4], seed=13)
    shards = [E(model, rank=r, nranks=nranks) for r in range(nranks)]
    infos = [s.layout() for s in shards]
    assert sum(i["n_owned"] for i in infos) == model.n
    assert sum(i["messages_per_sweep"] for i in infos) == model.messages_per_sweep()
    assert sum(i["algorithmic_bytes_per_sweep"] for i in infos) == model.algorithmic_bytes_per_sweep()
    assert len({i["segment_bytes"] for i in infos}) == 1
    slots = np.stack([s.node_slots() for s in shards])
    assert ((slots >= 0).sum(axis=0) == 1).all(), "each node is owned by exactly one rank"
    owner = (slots >= 0).argmax(axis=0)
    refs = [s.edge_refs() for s in shards]
    child = np.repeat(np.arange(model.n), np.diff(model.in_ptr))
    cut = owner[model.in_idx] != owner[child]
    assert sum(i["n_cut_edges"] for i in infos) == 2 * int(cut.sum())
    for e in np.nonzero(cut)[0]:
        a, b = owner[model.in_idx[e]], owner[child[e]]
        ga, gb = infos[a]["exchange_base"], infos[b]["exchange_base"]
        ra = (int(refs[a][0][e]) - ga, int(~refs[a][1][e]) - ga)
        rb = (int(refs[b][0][e]) - gb, int(~refs[b][1][e]) - gb)
        assert refs[a][1][e] < 0 and refs[b][1][e] < 0 and ra == rb and min(ra) >= 0   # same exchange slots from both ends
    for r in range(nranks):                                   # non-incident ranks hold no reference
        inc = (owner[model.in_idx] == r) | (owner[child] == r)
        assert (refs[r][0][~inc] == -1).all() and (refs[r][0][inc] >= 0).all()
    # exchange slots are disjoint: [start, start + chunks) intervals of all halves never overlap
    h = (model.k[model.in_idx] + 1) // 2
    starts, ends = [], []
    for e in np.nonzero(cut)[0]:
        a = owner[model.in_idx[e]]
        ga = infos[a]["exchange_base"]   # slots relative to the (rank-independent) exchange region
        pi, lam = int(refs[a][0][e]) - ga, int(~refs[a][1][e]) - ga
        starts += [pi, lam]
        ends += [pi + int(h[e]), lam + int(h[e])]
    order = np.argsort(starts)
    s, t = np.asarray(starts)[order], np.asarray(ends)[order]
    assert (s[1:] >= t[:-1]).all()


def test_default_partition_is_row_stripes_on_a_grid(E):
    g = synth.grid(64, 64, 4, seed=3)
    shards = [E(g, rank=r, nranks=8) for r in range(8)]
    owner = np.stack([s.node_slots() >= 0 for s in shards]).argmax(axis=0)
    assert (np.diff(owner) >= 0).all()                        # contiguous id ranges = row stripes
    counts = np.bincount(owner, minlength=8)
    assert counts.min() > 0.8 * g.n / 8 and counts.max() < 1.2 * g.n / 8


def test_evidence_helpers():
    m = synth.resume_chain()
    ev = Evidence.from_dict(m, {1: [0, 0, 1], 3: 0})
    assert ev.ne == 2 and ev.off.tolist() == [0, 3, 6] and ev.hard_states(m).tolist() == [-1, 2, -1, 0]
    with pytest.raises(ValueError):
        Evidence.from_dict(m, {2: [1.0, 0.0, 0.0]})
    with pytest.raises(ValueError):
        from_parent_lists([2, 2], [[1], []], [[.5, .5, .5, .5], [.5, .5]]).validate() or \
            from_parent_lists([2, 2], [[], [1]], [[.5, .5], [.5, .5, .5, .5]])


def test_sampler_mirror_loads_tables_and_files(tmp_path):
    """bn::sampler's host side (reference sampler.hpp:29-77, 166-190): table / file loading, counts."""
    from bayesiannetwork_amd.engine import Sampler
    smp = Sampler()
    assert smp.sampling_size() == 0 and smp.make_cpt(synth.pearl()) is False      # :83
    assert smp.load_sample({(0, 1, 1, 0): 3, (1, 1, 0, 0): 5})
    assert smp.sampling_size() == 8 and smp.table()[(1, 1, 0, 0)] == 5
    f = tmp_path / "samples.txt"
    f.write_text("3 0 1 1 0\n5   1 1 0 0\n2 0 1 1 0\n")                         # runs of blanks compress (:58)
    smp.set_filename(str(f))
    assert smp.sampling_size() == 0 and smp.table() == {} and smp.filename() == str(f)
    assert smp.load_sample([3, 2, 1, 0])                                           # column order = node list
    assert smp.sampling_size() == 10 and smp.table() == {(0, 1, 1, 0): 5, (0, 0, 1, 1): 5}
    smp.set_filename(str(tmp_path / "missing.txt"))
    assert smp.load_sample([0, 1, 2, 3]) is False                                  # :46


def test_any_arity_shapes_get_one_wavefront_per_node():
    """Shapes without a register-resident instantiation (mixed arities, arity > 4, > 64 CPT entries)
    are laid out for the flat variant: 8..64 lanes per node (the smallest group that holds the table at two
    entries per lane and each vector in one register); lanes_per_node = 1 keeps
    the one-lane-per-node generic path; more than 8 parents stays generic."""
    from bayesiannetwork_amd import _lib
    from bayesiannetwork_amd.engine import Engine
    m = synth.random_dag(120, 3, 16, [2, 3, 5, 4], seed=3)
    with Engine(m, device=_lib.BN_DEVICE_HOST_ONLY) as e:
        cls = e.layout_classes()
        assert any(c["variant"] == 3 for c in cls) and not any(c["variant"] == 0 for c in cls)
        for c in cls:
            if c["variant"] == 3:
                assert c["lanes_per_node"] in (8, 16, 32, 64)
        # every class packs 64 / lanes_per_node nodes into a tile (small tables share a wavefront)
        assert e.layout()["n_tiles"] == sum(-(-c["n_nodes"] // (64 // c["lanes_per_node"])) for c in cls)
        assert any(c["variant"] == 3 and c["lanes_per_node"] < 64 for c in cls)
    with Engine(m, device=_lib.BN_DEVICE_HOST_ONLY, lanes_per_node=1) as e:
        assert not any(c["variant"] in (2, 3) for c in e.layout_classes())
    big = synth.random_dag(40, 10, 20, 2, seed=4)   # up to 10 parents
    with Engine(big, device=_lib.BN_DEVICE_HOST_ONLY) as e:
        for c in e.layout_classes():
            assert (c["variant"] == 0) == (c["m"] > 8) or c["variant"] in (1, 2)


def test_mostly_any_arity_network_is_laid_out_uniformly():
    """When at least two thirds of the nodes only fit the any-arity variant, the templated shapes join
    it (the launch without register-resident tiles runs at twice the occupancy); every shard decides on
    the WHOLE model, so sharded and unsharded layouts agree."""
    from bayesiannetwork_amd import _lib
    from bayesiannetwork_amd.engine import Engine
    m = synth.random_dag(400, 3, 32, [5, 5, 5, 2, 5], seed=8)       # the k = 2 roots / chains are the minority
    with Engine(m, device=_lib.BN_DEVICE_HOST_ONLY) as e:
        assert {c["variant"] for c in e.layout_classes()} == {3}
    for r in range(2):
        with Engine(m, device=_lib.BN_DEVICE_HOST_ONLY, rank=r, nranks=2) as e:
            assert {c["variant"] for c in e.layout_classes()} == {3}
    g = synth.grid(6, 6, 4, seed=1)                                   # templated shapes only: unchanged
    with Engine(g, device=_lib.BN_DEVICE_HOST_ONLY) as e:
        assert {c["variant"] for c in e.layout_classes()} == {1}


def test_sharded_plan_puts_interior_tiles_first(bnlib):
    """Sharded plans order the tiles that touch no cut edge first (bn_plan.cpp): the overlapped run
    launches exactly those before the previous sweep's all-gather has landed, so none of their nodes may
    have a parent or a child on another rank -- and with one rank every tile is interior."""
    from bayesiannetwork_amd import _lib, synth
    from bayesiannetwork_amd.engine import Engine
    cases = [(synth.grid(40, 37, 4, seed=3), 4, None),
             (synth.random_dag(1500, 4, 48, [2, 3, 4], seed=12), 3, None)]
    g = synth.grid(24, 24, 3, seed=1)
    cases.append((g, 4, (synth.splitmix64(3, 0, g.n) % np.uint64(4)).astype(np.int32)))
    for model, world, owner_arg in cases:
        child = np.repeat(np.arange(model.n), np.diff(model.in_ptr))
        engines = [Engine(model, device=_lib.BN_DEVICE_HOST_ONLY, rank=r, nranks=world, owner=owner_arg) for r in range(world)]
        owner = np.full(model.n, -1, np.int32)
        for r, e in enumerate(engines):
            owner[e.node_slots() >= 0] = r
        assert (owner >= 0).all()
        cut_edge = owner[model.in_idx] != owner[child]
        touches = np.zeros(model.n, bool)
        touches[child[cut_edge]] = True
        touches[model.in_idx[cut_edge]] = True
        total_interior = 0
        for r, e in enumerate(engines):
            li, tiles = e.layout(), e.node_tiles()
            mine = owner == r
            assert (tiles[mine] >= 0).all() and (tiles[~mine] == -1).all()
            interior_nodes = mine & (tiles < li["n_interior_tiles"])
            assert not touches[interior_nodes].any(), "an interior tile holds a node with a cut edge"
            # every tile past the split does hold such a node
            late = np.unique(tiles[mine & (tiles >= li["n_interior_tiles"])])
            assert set(late.tolist()) == set(np.unique(tiles[mine & touches]).tolist())
            total_interior += li["n_interior_tiles"]
            e.close()
        if owner_arg is None:
            assert total_interior > 0
    with Engine(synth.grid(12, 12, 4, seed=2), device=_lib.BN_DEVICE_HOST_ONLY) as e:
        assert e.layout()["n_interior_tiles"] == e.layout()["n_tiles"]


def test_hot_kernels_do_not_spill(bnlib):
    """Code-object metadata of the built kernels (scripts/kernel_resources.py reads the gfx950 image
    embedded in csrc/*.o): the register-resident sweep instantiation has no spills, no scratch and no LDS;
    the resident kernel's lean instantiations (what grids and chains run) have no spills and hold their
    128 KiB of CPT halves in LDS, the all-shapes instantiations stay within a few dozen spilled dwords; the
    samplers do not spill."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "bayesiannetwork_amd", "csrc")
    need = ["bn_sweep_u.o", "bn_resident.o", "bn_lw_kernels.o", "bn_small.o", "bn_mid.o"]
    if not all(os.path.exists(os.path.join(csrc, f)) for f in need) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("object files / llvm tools not on this box")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(root, "scripts", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    u = kr.kernel_resources(os.path.join(csrc, "bn_sweep_u.o"))
    assert len(u) == 3  # plain stores, non-temporal stores, batched (one evidence set per blockIdx.y)
    for name, r in u.items():
        assert "bp_sweep_kernel" in name
        assert r["spill"] == 0 and r["scratch"] == 0 and r["lds"] == 0 and r["vgpr"] <= 256, (name, r)
    res = kr.kernel_resources(os.path.join(csrc, "bn_resident.o"))
    # {grid barrier, several sets, dataflow, dataflow of a shard} x {LEAN k = 2, 3, 4, all shapes at 8 waves per block;
    # k = 4 and all shapes at 4 waves per block, one per SIMD, with the whole register file}
    assert len(res) == 24
    import re
    for name, r in res.items():
        m = re.search(r"bp_resident_kernel<(\d+), (\d+), (\d+)>", name)
        assert m, name
        mode, lean, wmax = (int(x) for x in m.groups())
        if wmax == 4:           # one wave per SIMD: the whole register file (CPT entirely in registers: no LDS slots), nothing in scratch
            assert lean in (0, 4) and r["scratch"] == 0 and r["vgpr"] <= 512 and r["lds"] < 1024, (name, r)
            continue
        assert r["vgpr"] <= 256, (name, r)
        if mode == 3 and lean == 4:   # a shard's k = 4 tiles also carry the code for their cut edges: a handful of dwords
            assert r["spill"] <= 8, (name, r)
        elif lean != 0:         # LEAN = k: every node of arity k with <= 2 children -- what the headline grid runs (k = 4)
            assert r["spill"] == 0 and r["scratch"] == 0, (name, r)
        else:                   # every shape inlined into one kernel at two waves per SIMD (networks above ~900 tiles only)
            assert r["spill"] <= (96 if mode == 3 else 64), (name, r)
        if lean in (0, 4):      # 64-entry tables keep 36 entries per lane in LDS
            assert r["lds"] >= 144 * 1024, (name, r)
    for obj in ("bn_sweep_ug.o", "bn_sweep_all.o"):  # lane-group / any-arity instantiations: no spills either
        for name, r in kr.kernel_resources(os.path.join(csrc, obj)).items():
            assert r["spill"] == 0 and r["scratch"] == 0, (name, r)
    for name, r in kr.kernel_resources(os.path.join(csrc, "bn_lw_kernels.o")).items():
        assert r["spill"] == 0 and r["scratch"] == 0, (name, r)
    for name, r in kr.kernel_resources(os.path.join(csrc, "bn_mid.o")).items():   # the same items over several workgroups
        rounds = int(re.search(r"bp_mid_kernel<(\d+)>", name).group(1))
        assert r["vgpr"] <= 128 and (r["spill"] == 0 if rounds <= 2 else r["spill"] <= 32), (name, r)
    dag = kr.kernel_resources(os.path.join(csrc, "bn_dag.o"))   # register-resident DAG path: the single query, with the barrier and in its dataflow form
    for inst in ("bp_dag_kernel<false, false, false>", "bp_dag_kernel<false, false, true>"):
        hit = [r for name, r in dag.items() if inst in name]
        assert len(hit) == 1 and hit[0]["spill"] == 0 and hit[0]["scratch"] == 0 and hit[0]["vgpr"] <= 256, (inst, hit)
    small = kr.kernel_resources(os.path.join(csrc, "bn_small.o"))   # one workgroup per run, state in LDS (small networks)
    assert len(small) == 3
    for name, r in small.items():
        rounds = int(re.search(r"bp_small_kernel<(\d+)>", name).group(1))
        assert r["vgpr"] <= 128, (name, r)           # 16 waves per workgroup
        if rounds <= 2:                               # what ALARM-sized networks run
            assert r["spill"] == 0 and r["scratch"] == 0, (name, r)
        else:
            assert r["spill"] <= 16, (name, r)


def test_small_plan_emulated_equals_oracle(bnlib, oracle_mod):
    """The plan of the one-workgroup path (bn_small_plan.cpp), executed item by item on the CPU (tests/small_emulator.py),
    reproduces the oracle bit for bit -- beliefs, sweep count, residual history, final messages -- on networks with up to
    four parents per node, mixed arities, hard and soft evidence.  (On the GPU: tests/test_small_gpu.py.)"""
    import os
    import small_emulator
    from bayesiannetwork_amd import Evidence, _lib, engine, synth
    from bayesiannetwork_amd.dsc import load_dsc
    alarm, _ = load_dsc(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "alarm_shaped.dsc"))
    mixed = synth.random_dag(24, 4, 12, [2, 3, 4, 3, 2, 5], seed=12)
    soft = Evidence.from_dict(mixed, {3: np.full(int(mixed.k[3]), 1.0 / mixed.k[3]), 7: np.eye(int(mixed.k[7]))[0]})
    cases = [(synth.pearl(), Evidence.none(), 1e-9, 0), (alarm, synth.random_evidence(alarm, 0.1, seed=2), 1e-6, 0),
             (mixed, soft, 1e-6, 0), (mixed, synth.random_evidence(mixed, 0.1, seed=1), 1e-12, 7)]
    for g, ev, eps, cap in cases:
        with engine.Engine(g, device=_lib.BN_DEVICE_HOST_ONLY) as e:
            plan = e.small_plan()
        assert plan is not None
        got = small_emulator.emulate(plan, g, ev, eps, cap)
        want = oracle_mod.bp_run(g, ev, eps, cap, dump_msgs=True)
        assert got["sweeps"] == want["sweeps"]
        assert np.array_equal(got["beliefs"], want["beliefs"]) and np.array_equal(got["residuals"], want["residuals"])
        assert np.array_equal(got["pi_msg"], want["pi_msg"]) and np.array_equal(got["lambda_msg"], want["lambda_msg"])


def test_mid_plan_emulated_equals_oracle(bnlib, oracle_mod):
    """Networks beyond one workgroup's LDS: the plan that spreads the same items over several workgroups (bn_mid.hip), executed
    part by part on the CPU, reproduces the oracle bit for bit; the parts partition the nodes; the spill bounds of the kernel."""
    import small_emulator
    from bayesiannetwork_amd import _lib, engine, synth
    g = synth.random_dag(90, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=17)
    ev = synth.random_evidence(g, 0.08, seed=2)
    with engine.Engine(g, device=_lib.BN_DEVICE_HOST_ONLY) as e:
        assert e.info("small_eligible") == 0 and e.info("mid_eligible") == 1 and 2 <= e.info("mid_parts") <= 32
        parts = e.mid_plan()
    assert parts[0]["v0"] == 0 and parts[-1]["v1"] == g.n and all(a["v1"] == b["v0"] for a, b in zip(parts, parts[1:]))
    N, M = int(g.k.sum()), int(g.k[g.in_idx].sum())
    for kind, size in ((1, N), (2, M), (3, N), (4, M)):   # every output element has exactly one item, in exactly one part
        got = []
        for p in parts:
            slots = p["bslot"] if kind <= 2 else p["cslot"]
            got += (slots[(slots[:, 2] & 0xff) == kind][:, 1] & 0xffff).tolist()
        assert sorted(got) == list(range(size))
    got = small_emulator.emulate(parts, g, ev, 1e-6)
    want = oracle_mod.bp_run(g, ev, 1e-6, dump_msgs=True)
    assert got["sweeps"] == want["sweeps"] and np.array_equal(got["beliefs"], want["beliefs"]) and np.array_equal(got["residuals"], want["residuals"])
    assert np.array_equal(got["pi_msg"], want["pi_msg"]) and np.array_equal(got["lambda_msg"], want["lambda_msg"])
    with engine.Engine(synth.grid(64, 64, 4, seed=1), device=_lib.BN_DEVICE_HOST_ONLY) as e:   # well over 100 workgroups
        assert e.info("mid_eligible") == 1 and 100 <= e.info("mid_parts") <= 224
    with engine.Engine(synth.grid(128, 128, 4, seed=1), device=_lib.BN_DEVICE_HOST_ONLY) as e:   # more node-vector elements than the 16-bit indices hold
        assert e.info("mid_eligible") == 0 and e.mid_plan() is None


def test_dag_plan_emulated_equals_oracle(bnlib, oracle_mod):
    """Networks of arity <= 4 with up to 5 parents per node: the plan of the register-resident DAG path (bn_dag_plan.cpp), executed
    tile by tile and lane by lane on the CPU (tests/dag_emulator.py: the kernel's operation order, shuffle butterflies
    included).  Networks of nodes with <= 2 parents: the oracle bit for bit; lane-group tiles (factored contraction):
    <= 1e-12, equal sweep counts.  (On the GPU: tests/test_dag_gpu.py.)"""
    import dag_emulator
    from bayesiannetwork_amd import Evidence, _lib, engine, synth
    grid = synth.grid(12, 12, 4, seed=3)
    dag4 = synth.random_dag(300, 4, 32, 4, seed=5)
    dag5 = synth.random_dag(200, 5, 32, 4, seed=6)
    soft = Evidence.from_dict(dag4, {3: np.full(4, 0.25), 40: np.arange(1.0, 5.0), 70: 0})
    from helpers import hub_network
    hub20, hub70 = hub_network(20), hub_network(70)   # 21 parent items in one wave; more children than a wave has lanes
    cases = [(grid, synth.random_evidence(grid, 0.05, seed=1), 1e-6, 0, True), (grid, Evidence.none(), 1e-9, 0, True),
             (hub20, synth.random_evidence(hub20, 0.1, seed=1), 1e-9, 0, True), (hub70, synth.random_evidence(hub70, 0.1, seed=1), 1e-9, 0, True),
             (dag4, synth.random_evidence(dag4, 0.05, seed=2), 1e-6, 0, False), (dag4, soft, 1e-9, 0, False),
             (dag5, synth.random_evidence(dag5, 0.05, seed=3), 1e-6, 0, False), (dag5, Evidence.none(), 1e-12, 5, False)]
    # arities 2..4, padded to 4 (zeros where a state does not exist; the initial state in memory): the real entries keep the
    # reference's bits on networks of <= 2-parent nodes -- a zero term adds nothing to a sum -- and <= 1e-12 with lane groups
    mix2 = synth.random_dag(200, 2, 16, [2, 3, 4], seed=17)
    mixg = synth.grid(9, 9, 3, seed=4)
    mix4 = synth.random_dag(250, 4, 32, [2, 3, 4, 2], seed=18)
    softm = Evidence.from_dict(mix4, {v: np.linspace(0.2, 1.0, mix4.k[v]) for v in (5, 60, 200)})
    cases += [(mix2, synth.random_evidence(mix2, 0.1, seed=4), 1e-9, 0, True), (mix2, Evidence.none(), 1e-6, 0, True),
              (mixg, synth.random_evidence(mixg, 0.1, seed=5), 1e-9, 0, True),
              (mix4, synth.random_evidence(mix4, 0.05, seed=6), 1e-6, 0, False), (mix4, softm, 1e-9, 0, False), (mix4, Evidence.none(), 0.0, 4, False)]
    for g, ev, eps, cap, exact in cases:
        with engine.Engine(g, device=_lib.BN_DEVICE_HOST_ONLY) as e:
            assert e.info("dag_eligible") == 1
            plan = e.dag_plan()
        got = dag_emulator.emulate(plan, g, ev, eps, cap)
        want = oracle_mod.bp_run(g, ev, eps, cap, dump_msgs=True)
        assert got["sweeps"] == want["sweeps"]
        if exact:
            assert np.array_equal(got["beliefs"], want["beliefs"]) and np.array_equal(got["residuals"], want["residuals"])
            assert np.array_equal(got["pi_msg"], want["pi_msg"]) and np.array_equal(got["lambda_msg"], want["lambda_msg"])
        else:
            assert np.abs(got["beliefs"] - want["beliefs"]).max() < 1e-12 and np.abs(got["residuals"] - want["residuals"]).max() < 1e-12
            assert np.abs(got["pi_msg"] - want["pi_msg"]).max() < 1e-12 and np.abs(got["lambda_msg"] - want["lambda_msg"]).max() < 1e-12


def test_dag_plan_invariants(bnlib):
    """Every node sits in exactly one child tile (all lanes of its group), every edge and every node has exactly one parent
    item, the wave slots cover the tiles; BASELINE configs[1] fits the chip at one tile per wave, ten times the nodes do not
    (stream form); networks with another arity or more than 5 parents are not eligible; the kernel does not spill."""
    import importlib.util
    from bayesiannetwork_amd import _lib, engine, synth
    g = synth.random_dag(10000, 4, 64, 4, seed=1)
    with engine.Engine(g, device=_lib.BN_DEVICE_HOST_ONLY) as e:
        assert e.info("dag_eligible") == 1 and e.info("dag_stream") == 0 and e.info("mid_eligible") == 0
        p = e.dag_plan()
    assert p["n_tiles"] == p["child_tiles"] + p["parent_tiles"] <= p["blocks"] * 8 and p["blocks"] % 8 == 0 and p["blocks"] <= 224
    assert p["slot_ptr"][0] == 0 and p["slot_ptr"][-1] == p["n_tiles"] and (np.diff(p["slot_ptr"]) <= 1).all()
    kinds = p["tiles"][:, 0]
    m_of = np.diff(g.in_ptr)
    seen = np.zeros(g.n, dtype=np.int64)
    for t in np.nonzero(kinds <= 5)[0]:
        m = int(kinds[t])
        G = 1 if m <= 2 else 4 ** (m - 2)
        cn = p["cnode"][p["tiles"][t, 2]:p["tiles"][t, 2] + 64]
        nodes = cn[::G, 0]
        assert (cn[:, 0].reshape(-1, G) == nodes[:, None]).all()          # the lanes of a group share the node
        act = nodes[nodes >= 0]
        assert len(act) == p["tiles"][t, 1] and (m_of[act] == m).all() and (cn[::G, 1][nodes >= 0] == g.in_ptr[act]).all()
        seen[act] += 1
    assert (seen == 1).all()
    for b in p["tiles"][kinds == 8][:, 2]:   # a node's items sit in adjacent lanes of one wave, lambda(v) first, then the children ascending
        w = p["pitem"][b:b + 64]
        act = w[:, 0] >= 0
        rank = np.where(act, w[:, 3] >> 16, 0)
        first = np.arange(64) - (rank + 1)
        assert (first[act] >= 0).all() and (w[first[act], 0] == w[act, 0]).all() and ((w[first[act], 3] >> 16) == -1).all()
    assert (kinds == 9).sum() == 0
    it = np.concatenate([p["pitem"][b:b + 64] for b in p["tiles"][kinds == 8][:, 2]])
    it = it[it[:, 0] >= 0]
    assert sorted(it[it[:, 1] >= 0][:, 1].tolist()) == list(range(g.n_edges))      # one item per pi-message
    assert sorted(it[it[:, 1] < 0][:, 0].tolist()) == list(range(g.n))             # one per lambda(v)
    deg = np.bincount(g.in_idx, minlength=g.n)
    assert ((it[:, 3] & 0xffff) == deg[it[:, 0]]).all()
    msg = it[it[:, 1] >= 0]
    assert (p["oedge"][msg[:, 2] + (msg[:, 3] >> 16)] == msg[:, 1]).all()             # the target's rank among the node's children
    assert (g.in_idx[msg[:, 1]] == msg[:, 0]).all()
    big = synth.random_dag(30000, 4, 64, 4, seed=2)
    with engine.Engine(big, device=_lib.BN_DEVICE_HOST_ONLY) as e:
        assert e.info("dag_eligible") == 1 and e.info("dag_stream") == 1 and e.info("dag_blocks") == 224
        assert e.info("dag_tiles") > 224 * 8
    for other in (synth.random_dag(200, 4, 32, [4, 4, 5], seed=1), synth.random_dag(100, 6, 32, 4, seed=1)):   # an arity above 4; six parents
        with engine.Engine(other, device=_lib.BN_DEVICE_HOST_ONLY) as e:
            assert e.info("dag_eligible") == 0 and e.dag_plan() is None
    for padded in (synth.random_dag(200, 4, 32, [4, 4, 3], seed=1), synth.pearl()):   # arities below 4 are padded to 4
        with engine.Engine(padded, device=_lib.BN_DEVICE_HOST_ONLY) as e:
            assert e.info("dag_eligible") == 1 and e.dag_plan() is not None
    # out-degree bound (DagParentLane packs child count | target's rank << 16 into a signed word; a node with more than 63
    # children costs deg^2 record loads per sweep): 1 024 children are planned with every rank intact, 1 025 are refused
    from helpers import hub_network
    with engine.Engine(hub_network(1024), device=_lib.BN_DEVICE_HOST_ONLY) as e:
        assert e.info("dag_eligible") == 1
        ph = e.dag_plan()
    ih = ph["pitem"][ph["pitem"][:, 0] == 0]
    assert ih.shape[0] == 1025 and ((ih[:, 3] & 0xffff) == 1024).all()
    mh = ih[ih[:, 1] >= 0]
    assert np.array_equal(np.sort(mh[:, 3] >> 16), np.arange(1024)) and (ph["oedge"][mh[:, 2] + (mh[:, 3] >> 16)] == mh[:, 1]).all()
    with engine.Engine(hub_network(1025), device=_lib.BN_DEVICE_HOST_ONLY) as e:
        assert e.info("dag_eligible") == 0 and e.dag_plan() is None
    # image bound: the padded register image (4^(m+1) entries per node whatever the arities) stops at 256 MB -- a binary network
    # with 5-parent nodes is 64x its model there, and the stream form would re-read it every sweep
    wide = synth.random_dag(40000, 5, 32, 2, seed=3)
    assert sum(4 ** (m + 1) for m in np.diff(wide.in_ptr)) * 8 > 256 << 20
    with engine.Engine(wide, device=_lib.BN_DEVICE_HOST_ONLY) as e:
        assert e.info("dag_eligible") == 0
    obj = os.path.join(ROOT, "bayesiannetwork_amd", "csrc", "bn_dag.o")
    if os.path.exists(obj) and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "scripts", "kernel_resources.py"))
        kr = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(kr)
        res = {k: v for k, v in kr.kernel_resources(obj).items() if "bp_dag_kernel" in k}
        assert len(res) == 5          # {resident, stream form} x {a single query, several evidence sets per launch} + the single query's dataflow form
        for name, r in res.items():   # two waves per SIMD: 256 registers each, the 64 CPT entries of a lane among them
            assert r["vgpr"] <= 256, (name, r)
            if re.search(r"bp_dag_kernel<(true|false), false, (true|false)>", name):    # what a single query runs: nothing spilled
                assert r["spill"] == 0 and r["scratch"] == 0, (name, r)
            else:                     # the walk over a batch's sets: a handful of dwords
                assert r["spill"] <= 16, (name, r)


def test_reload_cpt_rebuilds_every_plan_host_only(bnlib, oracle_mod):
    """bn_reload_cpt on a host-only engine: the plans of the item kernels and of the register-resident DAG path carry the NEW
    tables afterwards (the emulators, run on them, give the oracle's result for the new network); a wrong length is refused.
    (The device images: tests/test_reload_gpu.py.)"""
    import dag_emulator
    import small_emulator
    from bayesiannetwork_amd import FlatModel, _lib, engine, synth
    from bayesiannetwork_amd.synth import _random_cpts
    for make, kind in ((lambda: synth.random_dag(24, 4, 12, [2, 3, 4, 3, 2, 5], seed=12), "small"),
                       (lambda: synth.random_dag(90, 3, 16, [2, 3, 4, 3, 2, 4, 5], seed=17), "mid"),
                       (lambda: synth.random_dag(300, 4, 32, 4, seed=5), "dag")):
        a = make()
        _, cpt = _random_cpts(a.k, a.in_ptr, a.in_idx, 99)
        b = FlatModel(a.k, a.in_ptr, a.in_idx, a.cpt_off, cpt)
        ev = synth.random_evidence(a, 0.08, seed=2)
        want = oracle_mod.bp_run(b, ev, 1e-6)
        with engine.Engine(a, device=_lib.BN_DEVICE_HOST_ONLY) as e:
            with pytest.raises(_lib.BnError, match="entries given"):
                e.reload_cpt(b.cpt[1:])
            e.reload_cpt(b.cpt)
            plan = {"small": e.small_plan, "mid": e.mid_plan, "dag": e.dag_plan}[kind]()
        assert plan is not None
        got = (dag_emulator if kind == "dag" else small_emulator).emulate(plan, b, ev, 1e-6)
        assert got["sweeps"] == want["sweeps"]
        if kind == "dag":
            assert np.abs(got["beliefs"] - want["beliefs"]).max() < 1e-12
        else:
            assert np.array_equal(got["beliefs"], want["beliefs"])


def test_dag_plan_is_light_until_asked_for(bnlib, oracle_mod):
    """bn_create builds the LIGHT plan of the register-resident DAG path (tile tables and the features the default-path policy reads);
    the padded CPT image is filled by the first use -- here bn_dag_plan_get.  A reload BEFORE that first use and one AFTER it both
    end in the plan a fresh engine builds from the new tables; the construction split is reported (bn_get_info "create_us_*")."""
    import dag_emulator
    from bayesiannetwork_amd import FlatModel, _lib, engine, synth
    from bayesiannetwork_amd.synth import _random_cpts
    a = synth.random_dag(400, 4, 32, [4, 3, 4, 2], seed=21)
    _, cpt = _random_cpts(a.k, a.in_ptr, a.in_idx, 5)
    b = FlatModel(a.k, a.in_ptr, a.in_idx, a.cpt_off, cpt)
    ev = synth.random_evidence(a, 0.05, seed=2)
    want = oracle_mod.bp_run(b, ev, 1e-6)
    with engine.Engine(b, device=_lib.BN_DEVICE_HOST_ONLY) as fresh:
        ref = fresh.dag_plan()
    for reload_first in (True, False):
        with engine.Engine(a, device=_lib.BN_DEVICE_HOST_ONLY) as e:
            assert e.info("dag_eligible") == 1 and e.info("dag_tiles") == ref["n_tiles"] and e.info("dag_blocks") == ref["blocks"]
            for k in ("plan", "small", "mid", "dag", "device"):
                assert e.info("create_us_" + k) >= 0
            assert e.info("create_us_device") == 0   # host-only: nothing was uploaded
            if not reload_first:
                assert e.dag_plan()["cpt_img"].size == ref["cpt_img"].size   # (first use: the full plan)
            e.reload_cpt(b.cpt)
            plan = e.dag_plan()
        for key in ("tiles", "slot_ptr", "cnode", "pitem", "oedge", "cpt_img", "npi_init"):
            assert np.array_equal(plan[key], ref[key]), (reload_first, key)
        got = dag_emulator.emulate(plan, b, ev, 1e-6)
        assert got["sweeps"] == want["sweeps"] and np.abs(got["beliefs"] - want["beliefs"]).max() < 1e-12
    with pytest.raises(_lib.BnError, match="unknown info"):
        engine.Engine(a, device=_lib.BN_DEVICE_HOST_ONLY).info("create_us_nothing")


def test_debug_stream_argument_checks(bnlib):
    """bn_debug_stream (the achievable-HBM yardstick of bench.py): arguments are checked before any device call; without a GPU it says so."""
    import ctypes
    from bayesiannetwork_amd import _lib
    g = ctypes.c_double(0.0)
    assert bnlib.bn_debug_stream(0, 0, 1 << 30, 3, None) == _lib.BN_ERR_ARG
    assert bnlib.bn_debug_stream(0, 2, 1 << 30, 3, ctypes.byref(g)) == _lib.BN_ERR_ARG
    assert bnlib.bn_debug_stream(0, 0, 1000, 3, ctypes.byref(g)) == _lib.BN_ERR_ARG
    assert bnlib.bn_debug_stream(0, 0, 1 << 30, 0, ctypes.byref(g)) == _lib.BN_ERR_ARG
    if not os.path.exists("/dev/kfd"):
        assert bnlib.bn_debug_stream(0, 0, 1 << 30, 3, ctypes.byref(g)) == _lib.BN_ERR_NO_DEVICE


def test_small_plan_invariants(bnlib):
    """Every output element has exactly one work item; a vector's elements sit in adjacent lanes of one wave; no two
    staged terms share a place and none lands in the zero padding of another run; the lanes of a wave add equally
    many terms; networks that do not fit say why."""
    from bayesiannetwork_amd import _lib, engine, synth
    g = synth.random_dag(40, 3, 16, [2, 3, 4, 3, 2, 4], seed=21)
    with engine.Engine(g, device=_lib.BN_DEVICE_HOST_ONLY) as e:
        p = e.small_plan()
        assert p is not None and e.info("small_lds_bytes") <= 150 * 1024 and 1 <= p["waves"] <= 16
    nt = 64 * p["waves"]
    N, M = int(g.k.sum()), int(g.k[g.in_idx].sum())
    assert (p["N"], p["M"], p["S"]) == (N, M, len(g.cpt))
    for slots, kinds, sizes in ((p["bslot"], (1, 2), (N, M)), (p["cslot"], (3, 4), (N, M))):
        for kind, size in zip(kinds, sizes):
            sel = slots[(slots[:, 2] & 0xff) == kind]
            assert sorted((sel[:, 1] & 0xffff).tolist()) == list(range(size))   # each element exactly once
        on = np.nonzero(slots[:, 2] & 0xff)[0]
        lane, first, k = on % 64, slots[on, 1] >> 24, (slots[on, 1] >> 16) & 0xff
        assert (lane >= first).all() and (first + k <= 64).all()                # a vector never leaves its wave
        at = lane - first
        assert ((slots[on - at, 1] & 0xffff) + at == (slots[on, 1] & 0xffff)).all()  # ... and its elements are adjacent
    b = p["bslot"].reshape(-1, 64, 4)
    assert all(len(set((row[:, 0] >> 16).tolist())) == 1 and (row[0, 0] >> 16) % 4 == 0 for row in b)
    # staged terms: one place each, inside the run of the accumulator that adds them
    valid = ((p["ent"][:, 1] >> 24) & 1) == 1
    assert valid.sum() == p["S"]
    places = (p["ent"][valid, 0] >> 16).tolist() + (p["term"] >> 16).tolist()
    assert len(set(places)) == len(places) and max(places) < p["T"]
    covered = set()
    for s_ in p["bslot"][(p["bslot"][:, 2] & 0xff) != 0]:
        base, n8 = int(s_[0]) & 0xffff, int(s_[0]) >> 16
        covered.update(range(base, base + n8))
    assert set(places) <= covered
    # not eligible: too large, more than 8 parents
    with engine.Engine(synth.grid(64, 64, 4, seed=1), device=_lib.BN_DEVICE_HOST_ONLY) as e:
        assert e.info("small_eligible") == 0 and e.small_plan() is None
    with engine.Engine(synth.random_dag(30, 9, 30, 2, seed=3), device=_lib.BN_DEVICE_HOST_ONLY) as e:
        assert e.info("small_eligible") == 0


def test_peer_blobs_and_flow_tables_host_only(bnlib):
    """bn_peer_export / bn_peer_import on host-only shard engines (no GPU): blob size and contents are consistent, the
    neighbour tables of all ranks mirror each other across the cut, a tile reports to exactly the ranks that hold one
    of its neighbours, and malformed input is an argument error, never a crash."""
    import ctypes
    from bayesiannetwork_amd import _lib, engine, synth
    g = synth.grid(30, 25, 4, seed=4)
    n = 4
    sh = [engine.Engine(g, device=_lib.BN_DEVICE_HOST_ONLY, rank=r, nranks=n) for r in range(n)]
    blobs = [s.peer_export() for s in sh]
    for s, b in zip(sh, blobs):
        assert len(b) == _lib.lib().bn_peer_blob_size(s._h) and s.info("n_boundary_nodes") * 8 < len(b)
        assert s.peer_import(blobs) is False          # host-only: tables only, no exchange set up
        assert s.info("nbr_chunks") >= 1 and s.info("shard_flow") == 0
    tabs = [s.flow_tables() for s in sh]
    tiles = [s.node_tiles() for s in sh]
    child = np.repeat(np.arange(g.n), np.diff(g.in_ptr))
    owner = np.full(g.n, -1)
    for r in range(n):
        owner[tiles[r] >= 0] = r
    assert (owner >= 0).all()
    for e_ in range(g.n_edges):                        # every cut edge: each side lists the other's tile and reports to its rank
        u, v = int(g.in_idx[e_]), int(child[e_])
        a, b = owner[u], owner[v]
        if a == b:
            continue
        tu, tv = int(tiles[a][u]), int(tiles[b][v])
        assert b * 2048 + tv in tabs[a][0][tu] and a * 2048 + tu in tabs[b][0][tv]
        assert (int(tabs[a][1][tu]) >> b) & 1 and (int(tabs[b][1][tv]) >> a) & 1
    for r in range(n):                                 # and nobody else
        nbr, pub = tabs[r]
        for t in range(nbr.shape[0]):
            ranks = {int(x) // 2048 for x in nbr[t][nbr[t] >= 0]} - {r}
            assert int(pub[t]) == sum(1 << q for q in ranks)
    with pytest.raises(_lib.BnError):                  # a blob of the wrong rank in slot 0
        sh[0].peer_import([blobs[1]] + blobs[1:])
    with pytest.raises(_lib.BnError):                  # too few blobs
        sh[0].peer_import(blobs[:2])
    with pytest.raises(_lib.BnError):                  # truncated blob
        sh[0].peer_import([blobs[0][:40]] + blobs[1:])
    other = synth.grid(30, 25, 4, seed=4)
    with engine.Engine(other, device=_lib.BN_DEVICE_HOST_ONLY) as one:   # not a sharded engine
        with pytest.raises(_lib.BnError):
            one.peer_export()
        assert one.info("flow_eligible") == 0 and one.info("resident_waves") in (4, 8)
        with pytest.raises(_lib.BnError):
            one.info("no_such_property")
    for s in sh:
        s.close()

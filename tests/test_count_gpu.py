"""HP-1 parity on a real MI355X: the HIP counting kernel (through the C ABI) against the golden vectors of the
reference and against the oracle on larger seeded inputs.  Integer work: bit-exact."""
import numpy as np
import pytest
import torch
import networkx as nx

from helpers import load, case_names, count_case

pytestmark = pytest.mark.gpu


def _global_ei(c):
    npt, ept, ei = c["node_ptr"], c["edge_ptr"], c["edge_index_local"].copy()
    for g in range(len(npt) - 1):
        ei[:, ept[g]:ept[g + 1]] += npt[g]
    return ei


@pytest.mark.parametrize("name", case_names("counts"))
def test_golden_counts(name):
    from gsn_amd.counting import CountPlan, count_batch
    c = count_case(name)
    plan = CountPlan.get(c["patterns"], c["mode"], c["induced"], c["directed_orbits"])
    out, st = count_batch(plan, c["node_ptr"], c["edge_ptr"], _global_ei(c), ids_are_global=True)
    assert out.dtype == torch.int64 and out.is_cuda
    assert (st.cpu().numpy() == 0).all()
    assert np.array_equal(out.cpu().numpy(), c["counts"])
    # graph-local ids, one graph at a time through the reference's per-graph signature
    out2, _ = count_batch(plan, c["node_ptr"], c["edge_ptr"], c["edge_index_local"], ids_are_global=False)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("name", case_names("counts_cliques"))
def test_golden_heavy_clique_counts(name):
    """the overflow-risk inputs of BASELINE config 3: K3..K5 on the IMDB-BINARY graphs with the most 5-cliques, bit-exact against
    tallies of networkx.enumerate_all_cliques (independent of VF2 and of the oracle)"""
    from gsn_amd.counting import CountPlan, count_batch
    c = count_case(name, "counts_cliques")
    plan = CountPlan.get(c["patterns"], c["mode"], c["induced"], c["directed_orbits"])
    out, st = count_batch(plan, c["node_ptr"], c["edge_ptr"], _global_ei(c), ids_are_global=True)
    assert (st.cpu().numpy() == 0).all()
    assert np.array_equal(out.cpu().numpy(), c["counts"])


@pytest.mark.parametrize("name", case_names("counts_stars"))
def test_golden_star8_counts(name):
    """star_graph(8): nine pattern vertices (GSN_KMAX 8 -> 9 in r06; --id_type star_graph --k 8, utils.py:59-62).  Small graphs against the
    reference's functions (VF2 stand-in), the heaviest IMDB-BINARY graphs against the closed form C(deg, 8) / sum C(deg - 1, 7) that
    make_golden.py pins to the reference on the small ones; vertex and edge mode, per-graph signatures included."""
    from gsn_amd.counting import CountPlan, count_batch
    c = count_case(name, "counts_stars")
    plan = CountPlan.get(c["patterns"], c["mode"], c["induced"], c["directed_orbits"])
    out, st = count_batch(plan, c["node_ptr"], c["edge_ptr"], _global_ei(c), ids_are_global=True)
    assert (st.cpu().numpy() == 0).all()
    assert np.array_equal(out.cpu().numpy(), c["counts"])
    out2, _ = count_batch(plan, c["node_ptr"], c["edge_ptr"], c["edge_index_local"], ids_are_global=False)
    assert torch.equal(out, out2)


def test_star8_orbits_and_ten_vertex_patterns_refused():
    from gsn_amd import _abi, patterns
    from gsn_amd.counting import CountPlan
    z = load("counts_stars")
    el = z["star8/edges"].tolist()
    _, part, memb, aut = patterns.automorphism_orbits(edge_list=el, print_msgs=False)
    assert [memb[v] for v in range(9)] == z["star8/v_membership"].tolist() and aut == 40320 and len(part) == 2
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        _, epart, ememb, _ = patterns.induced_edge_automorphism_orbits(edge_list=el)
    assert [ememb[i] for i in range(len(ememb))] == z["star8/e_membership"].tolist()
    with pytest.raises(_abi.GsnError, match="k <= 9"):
        CountPlan.get([list(nx.star_graph(9).edges)], "vertex", False)


def _cycles(ks):
    return [list(nx.cycle_graph(k).edges) for k in ks]


@pytest.mark.parametrize("mode,ks,induced", [("edge", range(3, 7), False), ("edge", range(3, 7), True),
                                             ("vertex", range(3, 9), False), ("vertex", range(3, 7), True)])
def test_zinc_batch_vs_oracle(mode, ks, induced):
    from gsn_amd import synth
    from gsn_amd.counting import counts2ids_batch
    from oracle import oracle
    b = synth.zinc_shape_batch(3000, seed=7)
    pats = _cycles(ks)
    got = counts2ids_batch(b, pats, mode, induced).cpu().numpy()
    local = b.edge_index - np.repeat(b.node_ptr[:-1], np.diff(b.edge_ptr))[None, :]
    ref = oracle.counts2ids(mode, induced, b.node_ptr, b.edge_ptr, local, pats, n_threads=8)
    assert np.array_equal(got, ref)
    assert got.sum() > 0


@pytest.mark.parametrize("mode", ["vertex", "edge"])
@pytest.mark.parametrize("induced", [False, True])
def test_er128_all_simple5_vs_oracle(mode, induced):
    """BASELINE config 5 shape: G(128, 1000), the 21 connected 5-vertex patterns (58 vertex / 56 edge orbit columns)."""
    from gsn_amd import synth
    from gsn_amd.counting import counts2ids_batch
    from oracle import oracle
    z = load("orbits")
    pats = [z["all_simple_graphs_5/%d/edges" % i].tolist() for i in range(21)]
    graphs = [synth.er_graph(128, 1000, s) for s in (0, 1)] + [synth.er_graph(100, 420, 2), synth.er_graph(128, 600, 3)]
    b = synth.collate(graphs)
    got = counts2ids_batch(b, pats, mode, induced).cpu().numpy()
    assert got.shape[1] == (58 if mode == "vertex" else 56)
    local = b.edge_index - np.repeat(b.node_ptr[:-1], np.diff(b.edge_ptr))[None, :]
    ref = oracle.counts2ids(mode, induced, b.node_ptr, b.edge_ptr, local, pats, n_threads=8)
    assert np.array_equal(got, ref)


def test_dense_cliques_vs_oracle():
    """IMDB-like: near-cliques with a hub, K3..K5, W=4 path (n up to 200)."""
    from gsn_amd import synth
    from gsn_amd.counting import counts2ids_batch
    from oracle import oracle
    rng = np.random.default_rng(3)
    graphs = []
    for n, p in ((40, 0.6), (136, 0.14), (200, 0.05), (64, 0.3), (65, 0.3)):
        a = np.triu(rng.random((n, n)) < p, 1)
        a[0, 1:] = True  # hub
        und = np.argwhere(a)
        graphs.append((n, synth.undirected_to_edge_index(n, und)))
    b = synth.collate(graphs)
    pats = [list(nx.complete_graph(k).edges) for k in (3, 4, 5)]
    for mode in ("vertex", "edge"):
        got = counts2ids_batch(b, pats, mode, False).cpu().numpy()
        local = b.edge_index - np.repeat(b.node_ptr[:-1], np.diff(b.edge_ptr))[None, :]
        ref = oracle.counts2ids(mode, False, b.node_ptr, b.edge_ptr, local, pats, n_threads=8)
        assert np.array_equal(got, ref)


def test_counts_beyond_16_bits_leave_the_lds_staging_exactly():
    """The output rows of a small graph are staged in LDS as 16-bit counts (count.hip, emit_cell); a count that does not fit leaves by
    itself.  Complete graphs give closed forms on both sides of 65 535 in ONE launch (and in one workgroup: K_20 + K_40 are a pair):
    a vertex of K_n lies in C(n - 1, k - 1) cliques K_k, an edge in C(n - 2, k - 2) (utils_graph_processing.py:103-179 counts each
    occurrence once per orbit position) -- K_40: C(39, 4) = 82 251 K_5 per vertex, C(39, 5) = 575 757 K_6; K_20: 3 876 / 11 628."""
    from math import comb
    from gsn_amd import synth
    from gsn_amd.counting import counts2ids_batch
    sizes = [20, 40, 40, 12, 33, 40]
    graphs = [(n, synth.undirected_to_edge_index(n, np.argwhere(np.triu(np.ones((n, n), dtype=bool), 1)))) for n in sizes]
    b = synth.collate(graphs)
    ks = (3, 4, 5, 6)
    pats = [list(nx.complete_graph(k).edges) for k in ks]
    v = counts2ids_batch(b, pats, "vertex", False).cpu().numpy()
    e = counts2ids_batch(b, pats, "edge", False).cpu().numpy()
    assert v.max() > 65535 and v.min() < 65535
    for g, n in enumerate(sizes):
        rows_v = v[b.node_ptr[g]:b.node_ptr[g + 1]]
        rows_e = e[b.edge_ptr[g]:b.edge_ptr[g + 1]]
        for c, k in enumerate(ks):
            assert (rows_v[:, c] == comb(n - 1, k - 1)).all(), (n, k, rows_v[0, c], comb(n - 1, k - 1))
            assert (rows_e[:, c] == comb(n - 2, k - 2)).all(), (n, k, rows_e[0, c], comb(n - 2, k - 2))


def test_vertex_edge_consistency_at_full_size():
    """Size-independent property at the ZINC-12k size of BASELINE config 2: for the cycle C_k,
    sum_v counts_v = k * #cycles and sum_e counts_e = 2k * #cycles, so both modes must agree; and the run is deterministic."""
    from gsn_amd import synth
    from gsn_amd.counting import counts2ids_batch
    b = synth.zinc_shape_batch(12000, seed=0)
    ks = list(range(3, 7))
    v = counts2ids_batch(b, _cycles(ks), "vertex", False)
    e = counts2ids_batch(b, _cycles(ks), "edge", False)
    e2 = counts2ids_batch(b, _cycles(ks), "edge", False)
    assert torch.equal(e, e2)
    for i, k in enumerate(ks):
        sv, se = int(v[:, i].sum()), int(e[:, i].sum())
        assert sv % k == 0 and se == 2 * sv
    assert int(v[:, 3].sum()) > 0  # six-rings exist
    # rows (u,v) and (v,u) carry the same counts (undirected orbits)
    ei = torch.from_numpy(b.edge_index).cuda()
    key = ei[0] * b.num_nodes + ei[1]
    rkey = ei[1] * b.num_nodes + ei[0]
    order, rorder = torch.argsort(key), torch.argsort(rkey)
    assert torch.equal(e[order], e[rorder])


def test_er128_full_size_vertex_edge_orbit_sums_agree():
    """Size-independent property at the ER size of BASELINE config 4 (G(128, 1000), the 21 connected five-vertex patterns) on
    10 240 graphs: every vertex-orbit column of a pattern sums to #occurrences * |orbit| and every edge-orbit column to
    #occurrences * (arcs in the class), so each column yields the same integer occurrence total per pattern -- per GRAPH, not only
    in the grand total; repeated runs are identical and the tail graphs equal a batch of their own (oracle-checked there)."""
    from gsn_amd import synth, patterns
    from gsn_amd.counting import counts2ids_batch
    from oracle import oracle
    z = load("orbits")
    pats = [z["all_simple_graphs_5/%d/edges" % i].tolist() for i in range(21)]
    n_graphs = 10240
    graphs = [synth.er_graph(128, 1000, s) for s in range(n_graphs)]
    b = synth.collate(graphs)
    v = counts2ids_batch(b, pats, "vertex", False)
    e = counts2ids_batch(b, pats, "edge", False)
    assert v.shape == (128 * n_graphs, 58) and e.shape == (2000 * n_graphs, 56)
    assert torch.equal(v, counts2ids_batch(b, pats, "vertex", False))
    vg = v.view(n_graphs, 128, 58).sum(1)                 # per graph column sums
    eg = e.view(n_graphs, 2000, 56).sum(1)
    cv = ce = 0
    nonzero = 0
    for el in pats:
        _, vpart, _, _ = patterns.automorphism_orbits(edge_list=el, print_msgs=False)
        _, epart, _, _ = patterns.induced_edge_automorphism_orbits(edge_list=el)
        occ = None
        for o in sorted(vpart):
            col = vg[:, cv]
            assert bool((col % len(vpart[o]) == 0).all())
            q = col // len(vpart[o])
            assert occ is None or torch.equal(q, occ)
            occ = q
            cv += 1
        for o in sorted(epart):
            col = eg[:, ce]
            assert bool((col % len(epart[o]) == 0).all())
            assert torch.equal(col // len(epart[o]), occ)
            ce += 1
        nonzero += int(occ.sum() > 0)
    assert cv == 58 and ce == 56 and nonzero == 21
    # the last three graphs as a batch of their own, against the oracle
    tail = synth.collate(graphs[-3:])
    tv = counts2ids_batch(tail, pats, "vertex", False)
    assert torch.equal(tv, v[-3 * 128:])
    local = tail.edge_index - np.repeat(tail.node_ptr[:-1], np.diff(tail.edge_ptr))[None, :]
    ref = oracle.counts2ids("vertex", False, tail.node_ptr, tail.edge_ptr, local, pats, n_threads=8)
    assert np.array_equal(tv.cpu().numpy(), ref)


@pytest.mark.parametrize("mode", ["vertex", "edge"])
@pytest.mark.parametrize("global_ids", [True, False])
def test_two_graphs_per_workgroup(mode, global_ids, capfd):
    """Batches of small graphs are counted two graphs per workgroup, as one disjoint union (csrc/count.hip): pairs that fit 64
    vertices, pairs that do not (redone one by one), an odd graph at the end, trailing isolated vertices in the FIRST graph of a pair
    (they are not vertices of the matched graph, utils_graph_processing.py:103-110), graphs without edges, duplicate columns and self
    loops, ids local to the graph and global; a bad index in the second graph of a pair is reported for that graph alone; a plan with
    a disconnected pattern takes the one-graph path.  Bit-exact against the oracle, and equal to the run with pairing off."""
    import os
    from gsn_amd import synth
    from gsn_amd.counting import CountPlan, count_batch
    from oracle import oracle
    rng = np.random.default_rng(11)
    graphs = [synth.zinc_shape_graph(rng) for _ in range(7)]
    graphs.append(synth.er_graph(40, 45, 3)); graphs.append(synth.er_graph(40, 40, 4))           # 80 vertices: the pair does not fit
    graphs.append((30, np.array([[0, 1, 1, 2, 2, 0, 3, 3], [1, 0, 2, 1, 0, 2, 3, 4]], dtype=np.int64)))   # triangle + self loop + an arc without its reverse; vertices 5..29 do not exist
    graphs.append(synth.zinc_shape_graph(rng))
    graphs.append((4, np.zeros((2, 0), dtype=np.int64))); graphs.append((3, np.zeros((2, 0), dtype=np.int64)))
    graphs.append(synth.er_graph(20, 45, 5))
    dup = np.array([[0, 1, 0, 1, 2, 1, 2, 0], [1, 0, 1, 0, 1, 2, 0, 2]], dtype=np.int64)
    graphs.append((3, dup))
    b = synth.collate(graphs)
    assert len(graphs) % 2 == 1
    pats = [list(nx.cycle_graph(k).edges) for k in (3, 4, 5)] + [list(nx.path_graph(3).edges), list(nx.star_graph(3).edges)]
    local = b.edge_index - np.repeat(b.node_ptr[:-1], np.diff(b.edge_ptr))[None, :]
    ei = b.edge_index if global_ids else local
    plan = CountPlan.get(pats, mode, False)

    def run(check=True, edge_index=ei):
        out, status = count_batch(plan, b.node_ptr, b.edge_ptr, torch.from_numpy(edge_index).cuda(), ids_are_global=global_ids, check=check)
        return out.cpu().numpy(), status.cpu().numpy()
    os.environ["GSN_CHAIN_TRACE"] = "1"
    try:
        got, st = run()
    finally:
        os.environ.pop("GSN_CHAIN_TRACE")
    assert "pair 1" in capfd.readouterr().err
    ref = oracle.counts2ids(mode, False, b.node_ptr, b.edge_ptr, local, pats, n_threads=4)
    assert np.array_equal(got, ref)
    assert not st.any()
    os.environ["GSN_COUNT_PAIR"] = "0"
    try:
        got1, _ = run()
    finally:
        os.environ.pop("GSN_COUNT_PAIR")
    assert np.array_equal(got, got1)
    # a bad index in the SECOND graph of the first pair: that graph's rows are zero and its status is raised, its neighbour's rows stay
    bad = ei.copy()
    c = int(b.edge_ptr[1])
    bad[0, c] = (b.node_ptr[2] if global_ids else b.node_ptr[2] - b.node_ptr[1]) + 5
    gotb, stb = run(check=False, edge_index=bad)
    rows = b.edge_ptr if mode == "edge" else b.node_ptr
    assert stb[1] != 0 and not stb[0] and not stb[2:].any()
    assert not gotb[rows[1]:rows[2]].any()
    assert np.array_equal(gotb[:rows[1]], ref[:rows[1]]) and np.array_equal(gotb[rows[2]:], ref[rows[2]:])
    # two triangles side by side: a disconnected pattern sees BOTH graphs of a union, so such a plan keeps one graph per workgroup
    if mode == "vertex":
        two = [(0, 1), (1, 2), (2, 0), (3, 4), (4, 5), (5, 3)]
        plan2 = CountPlan.get([two], mode, False)
        os.environ["GSN_CHAIN_TRACE"] = "1"
        try:
            out2, _ = count_batch(plan2, b.node_ptr, b.edge_ptr, torch.from_numpy(ei).cuda(), ids_are_global=global_ids)
        finally:
            os.environ.pop("GSN_CHAIN_TRACE")
        assert "pair 0" in capfd.readouterr().err
        ref2 = oracle.counts2ids(mode, False, b.node_ptr, b.edge_ptr, local, [two], n_threads=4)
        assert np.array_equal(out2.cpu().numpy(), ref2)


def test_reference_signatures_and_counts2ids():
    from gsn_amd import patterns, counting
    z = load("counts2ids")
    ptr, flat = z["pattern_ptr"], z["pattern_edges"]
    pats = [flat[ptr[i]:ptr[i + 1]].tolist() for i in range(len(ptr) - 1)]
    for mode in ("vertex", "edge"):
        fn_orb = patterns.automorphism_orbits if mode == "vertex" else patterns.induced_edge_automorphism_orbits
        dicts = []
        for el in pats:
            sg, part, memb, aut = fn_orb(edge_list=el, directed=False, directed_orbits=False)
            dicts.append({"subgraph": sg, "orbit_partition": part, "orbit_membership": memb, "aut_count": aut})
        cfn = counting.subgraph_isomorphism_vertex_counts if mode == "vertex" else counting.subgraph_isomorphism_edge_counts

        class Data:
            pass
        data = Data()
        ei = torch.from_numpy(z[mode + "/in_edge_index"])
        n = int(z[mode + "/in_num_nodes"])
        data.x = torch.ones(n, 1)
        data.edge_index = ei
        data.edge_features = torch.arange(ei.shape[1]) + 100
        res = counting.subgraph_counts2ids(cfn, data, dicts, {"induced": False, "directed": False})
        assert res is data
        assert torch.equal(res.edge_index, torch.from_numpy(z[mode + "/out_edge_index"]))
        assert torch.equal(res.edge_features, torch.from_numpy(z[mode + "/out_edge_features"]))
        assert res.identifiers.dtype == torch.int64 and not res.identifiers.is_cuda
        assert np.array_equal(res.identifiers.numpy(), z[mode + "/identifiers"])
        # per-pattern, per-graph call: CPU float64 like the reference (self loops passed straight in)
        c0 = cfn(ei, subgraph_dict=dicts[0], induced=False, num_nodes=n, directed=False)
        assert c0.dtype == torch.float64 and not c0.is_cuda
        ids_with_loops = np.zeros((ei.shape[1], 1)) if mode == "edge" else None
        if mode == "vertex":
            assert np.array_equal(c0.numpy()[:, 0], z[mode + "/identifiers"][:, 0])
        else:
            keep = (ei[0] != ei[1]).numpy()
            assert np.array_equal(c0.numpy()[keep, 0], z[mode + "/identifiers"][:, 0])
            assert (c0.numpy()[~keep] == 0).all()


def test_errors():
    from gsn_amd.counting import CountPlan, count_batch
    tri = [[(0, 1), (1, 2), (2, 0)]]
    plan = CountPlan.get(tri, "edge", False)
    ei = np.array([[0, 1, 1, 2, 2], [1, 0, 2, 1, 0]], dtype=np.int64)  # (0,2) missing but used by the triangle
    with pytest.raises(KeyError):
        count_batch(plan, [0, 3], [0, 5], ei, ids_are_global=False)
    # same graph, pattern that never touches the missing direction -> fine
    plan2 = CountPlan.get([[(0, 1), (1, 2), (2, 3), (3, 0)]], "edge", False)
    out, st = count_batch(plan2, [0, 3], [0, 5], ei, ids_are_global=False)
    assert int(out.sum()) == 0 and int(st[0]) == 0
    # under-declared sizes are reported, not silently wrong
    big = np.stack([np.arange(10), (np.arange(10) + 1) % 10]).astype(np.int64)
    big = np.concatenate([big, big[::-1]], axis=1)
    with pytest.raises(ValueError):
        count_batch(CountPlan.get(tri, "vertex", False), [0, 10], [0, 20], big, ids_are_global=False, max_nodes=4, max_edges=20)
    with pytest.raises(ValueError):   # id out of range
        count_batch(CountPlan.get(tri, "vertex", False), [0, 5], [0, 20], big, ids_are_global=False)
    # graph subset
    from gsn_amd import synth
    b = synth.zinc_shape_batch(10, seed=1)
    planv = CountPlan.get(tri + [[(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 0)]], "vertex", False)
    full, _ = count_batch(planv, b.node_ptr, b.edge_ptr, b.edge_index)
    part = torch.full_like(full, -7)
    count_batch(planv, b.node_ptr, b.edge_ptr, b.edge_index, graph_ids=[2, 5], out=part)
    for g in range(10):
        rows = slice(int(b.node_ptr[g]), int(b.node_ptr[g + 1]))
        if g in (2, 5):
            assert torch.equal(part[rows], full[rows])
        else:
            assert (part[rows] == -7).all()


def test_line_graph_orbits_in_the_edge_counter():
    """--edge_automorphism line_graph (deprecated): the reference's edge counter indexes the per-edge membership of
    edge_automorphism_orbits by directed-edge position and raises KeyError(m) on the first match
    (utils_graph_processing.py:161-173 with :241-243); graphs without a match get zero rows."""
    from gsn_amd import patterns
    from gsn_amd.counting import subgraph_isomorphism_edge_counts
    tri = [(0, 1), (1, 2), (2, 0)]
    g, part, memb, aut = patterns.edge_automorphism_orbits(edge_list=tri)
    d = {"subgraph": g, "orbit_partition": part, "orbit_membership": memb, "aut_count": aut}
    path = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])
    out = subgraph_isomorphism_edge_counts(path, subgraph_dict=d, induced=False)
    assert out.shape == (4, 1) and out.dtype == torch.float64 and float(out.abs().sum()) == 0.0
    tri_ei = torch.tensor([[0, 1, 1, 2, 2, 0], [1, 0, 2, 1, 0, 2]])
    with pytest.raises(KeyError) as e:
        subgraph_isomorphism_edge_counts(tri_ei, subgraph_dict=d, induced=False)
    assert e.value.args[0] == 3


def _large_graphs():
    """Graphs beyond 256 vertices (16-bit vertex ids, W = 8 / 12 words per adjacency row): PROTEINS-like sparse chains with
    rings (max 620 vertices in the TU dataset), a COLLAB-like union of cliques, and one mixed batch with small graphs."""
    from gsn_amd import synth
    rng = np.random.default_rng(17)
    out = []
    for n in (300, 620, 768):
        g = synth.zinc_shape_graph(rng, mean_n=n, sd_n=0.0, n_min=n, n_max=n, ring_rate=n / 14.0)
        out.append(g)
    n = 420
    und = set()
    for _ in range(9):
        size = int(rng.integers(8, 22))
        mem = rng.choice(n, size=size, replace=False)
        for i in range(size):
            for j in range(i + 1, size):
                und.add((int(min(mem[i], mem[j])), int(max(mem[i], mem[j]))))
    out.append((n, synth.undirected_to_edge_index(n, sorted(und))))
    out.append(synth.er_graph(40, 90, 5))
    return out


@pytest.mark.parametrize("mode,pats,induced", [("vertex", "cycles", False), ("vertex", "cliques", False), ("edge", "cycles", True),
                                               ("vertex", "cycles", True), ("edge", "cliques", False)])
def test_large_graphs_vs_oracle(mode, pats, induced):
    from gsn_amd import synth
    from gsn_amd.counting import counts2ids_batch
    from oracle import oracle
    plist = _cycles(range(3, 7)) if pats == "cycles" else [list(nx.complete_graph(k).edges) for k in (3, 4, 5)]
    b = synth.collate(_large_graphs())
    got = counts2ids_batch(b, plist, mode, induced).cpu().numpy()
    local = b.edge_index - np.repeat(b.node_ptr[:-1], np.diff(b.edge_ptr))[None, :]
    ref = oracle.counts2ids(mode, induced, b.node_ptr, b.edge_ptr, local, plist, n_threads=8)
    assert np.array_equal(got, ref)
    assert got.sum() > 0


def test_too_large_graph_is_refused():
    from gsn_amd import _abi, synth
    from gsn_amd.counting import counts2ids_batch
    b = synth.collate([synth.er_graph(800, 1000, 1)])
    with pytest.raises(_abi.GsnError, match="768"):
        counts2ids_batch(b, _cycles([3]), "vertex", False)


@pytest.mark.parametrize("mode", ["vertex", "edge"])
@pytest.mark.parametrize("induced", [False, True])
def test_all_six_vertex_patterns_vs_oracle(mode, induced):
    """--id_type all_simple_graphs --k 6: the 112 connected six-vertex patterns (their orbit tables are pinned to the
    reference in orbits.npz) counted on small random graphs, against the oracle."""
    from gsn_amd import synth
    from gsn_amd.counting import counts2ids_batch
    from oracle import oracle
    z = load("orbits")
    pats = [z["all_simple_graphs_6/%d/edges" % i].tolist() for i in range(112)]
    graphs = [synth.er_graph(14, 30, 1), synth.er_graph(18, 40, 2), synth.er_graph(12, 40, 3), synth.er_graph(20, 30, 4)]
    b = synth.collate(graphs)
    got = counts2ids_batch(b, pats, mode, induced).cpu().numpy()
    local = b.edge_index - np.repeat(b.node_ptr[:-1], np.diff(b.edge_ptr))[None, :]
    ref = oracle.counts2ids(mode, induced, b.node_ptr, b.edge_ptr, local, pats, n_threads=8)
    assert got.shape == ref.shape and got.shape[1] > 300
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("k", [7, 8])
def test_sampled_seven_and_eight_vertex_patterns_vs_oracle(k):
    """k = 7 (a sample of the 853 connected atlas graphs) and k = 8 (random connected graphs): plan compiler incl.
    symmetry breaking, distance constraints and 8-level searches, vertex and edge mode."""
    from gsn_amd import synth
    from gsn_amd.counting import counts2ids_batch
    from oracle import oracle
    rng = np.random.default_rng(k)
    if k == 7:
        from networkx.generators.atlas import graph_atlas_g
        pool = [g for g in graph_atlas_g() if g.number_of_nodes() == 7 and nx.is_connected(g)]
        pats = [list(pool[i].edges) for i in rng.choice(len(pool), size=40, replace=False)]
    else:
        pats = []
        while len(pats) < 16:
            g = nx.gnm_random_graph(8, int(rng.integers(7, 18)), seed=int(rng.integers(1 << 30)))
            if nx.is_connected(g):
                pats.append(list(g.edges))
    b = synth.collate([synth.er_graph(13, 30, 11), synth.er_graph(16, 34, 12)])
    local = b.edge_index - np.repeat(b.node_ptr[:-1], np.diff(b.edge_ptr))[None, :]
    for mode in ("vertex", "edge"):
        got = counts2ids_batch(b, pats, mode, False).cpu().numpy()
        ref = oracle.counts2ids(mode, False, b.node_ptr, b.edge_ptr, local, pats, n_threads=8)
        assert np.array_equal(got, ref), mode


@pytest.mark.parametrize("mode", ["vertex", "edge"])
def test_count_with_fused_encoding(mode):
    """gsn_count_encode_hip: the one-hot rows written by the counting kernel equal gsn_one_hot_hip / torch one_hot of the int64
    counts -- clamped and unclamped, mixed class counts, with and without the int64 rows, graph subsets, a refused graph."""
    from gsn_amd import synth, layers
    from gsn_amd.counting import CountPlan, count_batch
    b = synth.zinc_shape_batch(2000, seed=21)
    pats = _cycles(range(3, 7))
    plan = CountPlan.get(pats, mode, False)
    ref, _ = count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index)
    for n_classes, clamp in (([3, 3, 3, 3], True), ([2, 5, 1, 4], True), ([2, 2, 3, 2], False)):
        want = layers.one_hot_identifiers(ref, n_classes, clamp=clamp)
        out, st, enc = count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index, encode=(n_classes, clamp))
        assert torch.equal(out, ref) and torch.equal(enc, want) and enc.dtype == torch.float32
        out2, st, enc2 = count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index, encode=(n_classes, clamp), counts=False)
        assert out2 is None and torch.equal(enc2, want)
    # a subset of the graphs into a reused buffer: the other rows are left alone
    buf = torch.full((ref.shape[0], 12), -5.0, device="cuda")
    count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index, graph_ids=[3, 1000], encode=([3, 3, 3, 3], True), counts=False, encoded_out=buf)
    want = layers.one_hot_identifiers(ref, [3, 3, 3, 3], clamp=True)
    ptr = b.edge_ptr if mode == "edge" else b.node_ptr
    for g in (2, 3, 999, 1000, 1001):
        rows = slice(int(ptr[g]), int(ptr[g + 1]))
        assert torch.equal(buf[rows], want[rows]) if g in (3, 1000) else bool((buf[rows] == -5.0).all())
    # a graph beyond the declared sizes: zero rows + status, like the int64 path
    with pytest.raises(ValueError):
        count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index, max_nodes=8, max_edges=200, encode=([3, 3, 3, 3], True))
    with pytest.raises(ValueError):
        count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index, encode=([3, 3, 3], True))


@pytest.mark.parametrize("mode", ["vertex", "edge"])
def test_fused_encoding_on_split_graphs(mode):
    """Few heavy graphs: several workgroups share one graph's cells (split > 1), so the class indices cannot be staged per graph
    and every cell writes its floats itself; duplicates and self loops in the columns."""
    from gsn_amd import synth, layers
    from gsn_amd.counting import CountPlan, count_batch
    graphs = [synth.er_graph(128, 900, s) for s in (5, 6, 7)]
    n0, e0 = graphs[0]
    graphs[0] = (n0, np.concatenate([e0, e0[:, :7], np.array([[3, 9], [3, 9]])], axis=1))      # repeated columns + self loops
    b = synth.collate(graphs)
    plan = CountPlan.get(_cycles(range(3, 6)), mode, False)
    ref, _ = count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index)
    n_classes = [4, 7, 250]
    for clamp in (True, False):
        want = layers.one_hot_identifiers(ref, n_classes, clamp=clamp)
        out, _, enc = count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index, encode=(n_classes, clamp))
        assert torch.equal(out, ref) and torch.equal(enc, want)
        _, _, enc2 = count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index, encode=(n_classes, clamp), counts=False)
        assert torch.equal(enc2, want)
    assert int(ref.max()) > 7          # (the clamp matters)


PENDANT_PATTERNS = {
    "star4": [(0, 1), (0, 2), (0, 3), (0, 4)],                       # four twin leaves
    "path5": [(0, 1), (1, 2), (2, 3), (3, 4)],                       # chain tail from an end, independent ends from the middle
    "fork": [(0, 1), (1, 2), (2, 3), (2, 4)],                        # two twin leaves on the far end of a path
    "tadpole32": [(0, 1), (1, 2), (2, 0), (2, 3), (3, 4)],           # triangle with a tail of two
    "c4_pendant": [(0, 1), (1, 2), (2, 3), (3, 0), (0, 4)],
    "bull": [(0, 1), (1, 2), (2, 0), (0, 3), (1, 4)],                # two independent pendants on a triangle
    "spider6": [(0, 1), (0, 2), (0, 3), (1, 4), (2, 5)],             # six vertices: pendants at different depths
    "star5": [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5)],               # runs of twin leaves: C(n, r), r up to 5
    "star6": [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6)],
    "broom": [(0, 1), (1, 2), (2, 3), (2, 4), (2, 5), (2, 6)],
}


@pytest.mark.parametrize("mode", ["vertex", "edge"])
@pytest.mark.parametrize("induced", [False, True])
def test_closed_form_tails_vs_oracle(mode, induced):
    """The last two levels of a rooted search in closed form / in one tight loop (count_core.h: tail_pairs, tail_loop; round 3): patterns
    with pendant vertices on molecule-sized graphs (one bit-matrix word: the TAIL instantiation of the one-wave kernel), on ER graphs
    of 100-128 vertices (two words: closed forms + the tight loop, one frame fewer) and of 200 vertices (four words), non-induced (closed
    forms apply) and induced (they do not: same counts through the generic levels / the loop) -- bit-exact against the oracle.  The plan
    table of the non-induced run holds every closed-form kind."""
    from gsn_amd import synth
    from gsn_amd.counting import CountPlan, counts2ids_batch
    from oracle import oracle
    pats = [PENDANT_PATTERNS[k] for k in sorted(PENDANT_PATTERNS)]
    if not induced:
        plan = CountPlan.get(pats, mode, False)
        arr = next(v for v in (getattr(plan, n) for n in dir(plan)) if isinstance(v, np.ndarray) and v.dtype == np.uint32)
        n_plans, plans_off = int(arr[3]), int(arr[7])
        from gsn_amd.counting import PLAN_STRIDE_WORDS as PS
        kinds = {(int(arr[plans_off + i * PS + 1]) >> 28) & 3 for i in range(n_plans)}
        assert kinds == {0, 1, 2, 3}, kinds
    # (seven-vertex stars: the oracle enumerates every map, d! / (d - 6)! per vertex of degree d -- moderate degrees)
    batches = [synth.zinc_shape_batch(40, seed=5),
               synth.collate([synth.er_graph(128, 420, 11), synth.er_graph(100, 300, 12), synth.er_graph(90, 200, 13), synth.er_graph(65, 64, 14)]),
               synth.collate([synth.er_graph(200, 600, 21), synth.er_graph(150, 300, 22)])]
    for b in batches:
        got = counts2ids_batch(b, pats, mode, induced).cpu().numpy()
        local = b.edge_index - np.repeat(b.node_ptr[:-1], np.diff(b.edge_ptr))[None, :]
        ref = oracle.counts2ids(mode, induced, b.node_ptr, b.edge_ptr, local, pats, n_threads=8)
        assert np.array_equal(got, ref), (mode, induced, b.num_nodes)

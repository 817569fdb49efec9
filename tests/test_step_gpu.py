"""The counting kernel's side outputs (gsn_count_encode_pack16_side_hip) and HP-1 + HP-2 as one host call (gsn_count_layer_step_hip,
gsn_amd.step.CountLayerStep) against the separate entry points they replace: the CSR of gsn_csr_build_graphs_hip (GSN_sparse.py:140-143),
the packs of gsn_one_hot_pack16_hip / gsn_count_encode_pack16_hip (utils_graph_learning.py:170-187), the identifiers of gsn_count_hip
(utils_ids.py:7-29) and the layer rows of ``layer(Codes, ...)`` (GSN_edge_sparse.py:82-170) -- all bit for bit."""
import networkx as nx
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
import os
no_cache = pytest.mark.skipif(os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") == "1",
                              reason="stream capture cannot free memory without the caching allocator (scripts/oob_check.sh)")

CTOR = dict(d_in=28, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=128,
            d_up=128, d_h=[128], seed=0, activation_name="relu", bn=True, msg_kind="general")


def _dev():
    return torch.device("cuda", 0)


def _zinc(n_graphs, seed):
    from gsn_amd import synth
    b = synth.zinc_shape_batch(n_graphs, seed=seed)
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return b, t(b.node_ptr), t(b.edge_ptr), t(b.edge_index), t(b.atom_type), t(b.bond_type)


def _cycles(ks=range(3, 7)):
    return [list(nx.cycle_graph(k).edges) for k in ks]


def _reference_csr(ei, row, N, node_ptr, edge_ptr, max_nodes, max_edges):
    from gsn_amd._index import build_csr_graphs
    return build_csr_graphs(ei[row], N, node_ptr, edge_ptr, max_nodes, max_edges, other=ei[1 - row], check=True)


@pytest.mark.parametrize("n_graphs,csr_row", [(1, 1), (2, 1), (7, 0), (257, 1), (1000, 0)])
def test_side_outputs_equal_the_separate_launches_on_molecules(n_graphs, csr_row):
    from gsn_amd import layers, packs
    from gsn_amd.counting import CountPlan, count_batch, count_batch_side
    b, node_ptr, edge_ptr, ei, atoms, bonds = _zinc(n_graphs, 100 + n_graphs)
    N, E = b.num_nodes, b.num_edges
    mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
    plan = CountPlan.get(_cycles(), "edge", False)
    xc, efc = layers.Codes(atoms, [28]), layers.Codes(bonds, [4])
    r = count_batch_side(plan, node_ptr, edge_ptr, ei, mn, me, id_classes=[3, 3, 3, 3], clamp=True, x_codes=xc, ef_codes=efc, csr_row=csr_row)
    ids_ref, st = count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=_dev())
    assert torch.equal(r["ids"], ids_ref)
    assert int(r["status"].abs().sum()) == 0 and int(r["code_status"].item()) == 0
    seg, perm, tgt, src = _reference_csr(ei, csr_row, N, node_ptr, edge_ptr, mn, me)
    c = r["csr"]
    assert torch.equal(c.seg_ptr, seg) and torch.equal(c.perm, perm) and torch.equal(c.tgt, tgt) and torch.equal(c.src, src)
    npk = packs.pack_node_codes(layers.Codes(atoms, [28]))
    assert torch.equal(r["node_pack"].view(torch.int16), npk.view(torch.int16))
    epk = packs.new_edge_pack(E, _dev())
    count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=_dev(), encode=([3, 3, 3, 3], True),
                encoded_pack=(epk, 0), encoded_rows=False)
    packs.pack_edge_codes(layers.Codes(bonds, [4]), epk, 12)
    assert torch.equal(r["edge_pack"].view(torch.int16), epk.view(torch.int16))


def test_side_csr_is_stable_on_multigraph_columns_and_self_loops():
    """Duplicate columns and self loops: the CSR keeps every column (stable by column id inside a target), the counts see the simple graph."""
    from gsn_amd import layers
    from gsn_amd.counting import CountPlan, count_batch, count_batch_side
    rng = np.random.default_rng(5)
    node_ptr, edge_ptr, cols = [0], [0], []
    for g in range(41):
        n = int(rng.integers(2, 30))
        m = int(rng.integers(0, 40))
        u = rng.integers(0, n, m); v = rng.integers(0, n, m)         # undirected pairs, self loops and repeats included
        both = np.concatenate([np.stack([u, v]), np.stack([v, u])], 1)   # both directions (a missing direction is the reference's KeyError)
        both = both[:, rng.permutation(both.shape[1])]                # columns in no particular order
        cols.append(both + node_ptr[-1])
        node_ptr.append(node_ptr[-1] + n); edge_ptr.append(edge_ptr[-1] + both.shape[1])
    ei_np = np.ascontiguousarray(np.concatenate(cols, 1).astype(np.int64))
    dev = _dev()
    node_ptr_t, edge_ptr_t, ei = (torch.tensor(a, dtype=torch.int64, device=dev) for a in (node_ptr, edge_ptr, ei_np))
    N, E = node_ptr[-1], edge_ptr[-1]
    mn, me = int(np.diff(node_ptr).max()), int(np.diff(edge_ptr).max())
    plan = CountPlan.get(_cycles(range(3, 6)), "edge", False)          # three columns: the generic instantiation
    atoms = torch.from_numpy(rng.integers(0, 9, (N, 2))).to(dev)
    xc = layers.Codes(atoms, [5, 9], clamp=True)                        # codes 5..8 of the first column are clamped to class 4
    for row in (0, 1):
        r = count_batch_side(plan, node_ptr_t, edge_ptr_t, ei, mn, me, x_codes=xc, csr_row=row, register=False)
        seg, perm, tgt, src = _reference_csr(ei, row, N, node_ptr_t, edge_ptr_t, mn, me)
        c = r["csr"]
        assert torch.equal(c.seg_ptr, seg) and torch.equal(c.perm, perm) and torch.equal(c.tgt, tgt) and torch.equal(c.src, src)
        # (the stable sort by target, directly)
        order = np.argsort(ei_np[row], kind="stable")
        assert np.array_equal(c.perm.cpu().numpy(), order)
        ids_ref, _ = count_batch(plan, node_ptr_t, edge_ptr_t, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, check=True)
        assert torch.equal(r["ids"], ids_ref)
        from gsn_amd import packs
        npk = packs.pack_node_codes(layers.Codes(atoms, [5, 9], clamp=True))
        assert torch.equal(r["node_pack"].view(torch.int16), npk.view(torch.int16))


@pytest.mark.parametrize("mode", ["vertex", "edge"])
def test_side_csr_and_node_pack_on_larger_graphs(mode):
    """ER graphs of 100-128 vertices (two-word adjacency rows, 256-thread workgroups): CSR + node pack beside plain counts, no identifier pack."""
    from gsn_amd import layers, packs, synth
    from gsn_amd.counting import CountPlan, count_batch, count_batch_side
    rng = np.random.default_rng(11)
    graphs = [synth.er_graph(int(rng.integers(100, 129)), 200, seed=50 + i) for i in range(2100)]      # (>= 2048 graphs: one workgroup per graph)
    b = synth.collate(graphs)
    dev = _dev()
    node_ptr, edge_ptr, ei = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (b.node_ptr, b.edge_ptr, b.edge_index))
    N = b.num_nodes
    mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
    plan = CountPlan.get([list(nx.cycle_graph(3).edges), list(nx.path_graph(4).edges)], mode, False)
    codes = torch.from_numpy(rng.integers(0, 7, (N, 1))).to(dev)
    xc = layers.Codes(codes, [6])                                       # code 6 is out of range: a zero block and the status flag
    r = count_batch_side(plan, node_ptr, edge_ptr, ei, mn, me, x_codes=xc, csr_row=1, register=False)
    ids_ref, _ = count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev)
    assert torch.equal(r["ids"], ids_ref)
    seg, perm, tgt, src = _reference_csr(ei, 1, N, node_ptr, edge_ptr, mn, me)
    c = r["csr"]
    assert torch.equal(c.seg_ptr, seg) and torch.equal(c.perm, perm) and torch.equal(c.tgt, tgt) and torch.equal(c.src, src)
    assert int(r["code_status"].item()) == 1
    npk = packs.new_node_pack(N, dev)
    packs._pack_codes(layers.Codes(codes, [6]), npk, 0, packs.NODE_COLS - 1, check=False)
    assert torch.equal(r["node_pack"].view(torch.int16), npk.view(torch.int16))


def test_a_split_launch_refuses_the_side_outputs():
    """Few heavy graphs are split over several workgroups each: no single workgroup owns a graph -> GSN_E_UNSUPPORTED, nothing launched."""
    from gsn_amd import _abi, synth
    from gsn_amd.counting import CountPlan, count_batch_side
    b = synth.collate([synth.er_graph(120, 300, seed=i) for i in range(5)])
    dev = _dev()
    node_ptr, edge_ptr, ei = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (b.node_ptr, b.edge_ptr, b.edge_index))
    plan = CountPlan.get([list(nx.cycle_graph(3).edges), list(nx.path_graph(4).edges)], "edge", False)
    with pytest.raises(_abi.GsnError, match="side workgroups"):
        count_batch_side(plan, node_ptr, edge_ptr, ei, 120, 600, csr_row=1, register=False)


def test_under_declared_sizes_raise_the_status_and_keep_the_csr_in_bounds():
    """Graphs beyond the declared max_nodes: GSN_ST_TOO_LARGE and zero identifiers (a pair of small graphs that fits together is still counted);
    the CSR stays a permutation inside the batch, every row of both packs is encoded all the same."""
    from gsn_amd import layers
    from gsn_amd.counting import CountPlan, count_batch, count_batch_side
    b, node_ptr, edge_ptr, ei, atoms, bonds = _zinc(33, 3)
    N, E = b.num_nodes, b.num_edges
    sizes = np.diff(b.node_ptr)
    mn, me = 12, int(np.diff(b.edge_ptr).max())
    plan = CountPlan.get(_cycles(), "edge", False)
    r = count_batch_side(plan, node_ptr, edge_ptr, ei, mn, me, id_classes=[3, 3, 3, 3], x_codes=layers.Codes(atoms, [28]),
                         ef_codes=layers.Codes(bonds, [4]), csr_row=1, register=False)
    st = r["status"].cpu().numpy()
    assert set(np.unique(st)) <= {0, 2} and (st[sizes > 2 * mn] == 2).all() and (sizes[st == 2] > mn).all() and (st == 2).any()
    seg = r["csr"].seg_ptr.cpu().numpy()
    assert seg[0] == 0 and seg[-1] == E and (np.diff(seg) >= 0).all()
    perm = r["csr"].perm.cpu().numpy()
    assert np.array_equal(np.sort(perm), np.arange(E))
    tgt = r["csr"].tgt.cpu().numpy()
    assert np.array_equal(tgt, b.edge_index[1][perm]) and (np.diff(tgt) >= 0).all()      # (these graphs fit the launch's LDS: sorted all the same)
    ep = r["edge_pack"].float().cpu().numpy()
    assert np.array_equal(ep[:, 12:].argmax(1), b.bond_type) and (ep[:, 12:].sum(1) == 1).all()
    npk = r["node_pack"].float().cpu().numpy()
    assert np.array_equal(npk[:, :28].argmax(1), b.atom_type) and (npk[:, :28].sum(1) == 1).all() and (npk[:, 31] == 1).all() and (npk[:, 28:31] == 0).all()
    ids_ref, _ = count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, device=_dev())
    ids = r["ids"].cpu().numpy()
    for g in range(b.num_graphs):
        rows = slice(int(b.edge_ptr[g]), int(b.edge_ptr[g + 1]))
        if st[g] == 2:
            assert (ids[rows] == 0).all() and (ep[rows, :12] == 0).all()
        else:
            assert np.array_equal(ids[rows], ids_ref[rows].cpu().numpy())


@pytest.mark.parametrize("flow", ["source_to_target", "target_to_source"])
@pytest.mark.parametrize("n_graphs", [1, 64, 999])
def test_one_call_step_equals_counting_then_the_layer(flow, n_graphs):
    from gsn_amd import layers, packs
    from gsn_amd.counting import CountPlan, count_batch
    from gsn_amd.step import CountLayerStep
    b, node_ptr, edge_ptr, ei, atoms, bonds = _zinc(n_graphs, 7 + n_graphs)
    N, E = b.num_nodes, b.num_edges
    mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
    dev = _dev()
    plan = CountPlan.get(_cycles(), "edge", False)
    torch.manual_seed(0)
    layer = layers.GSN_edge_sparse(flow=flow, **CTOR).to(dev).eval()
    with torch.no_grad():                                               # running statistics that are not the identity
        for m in layer.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    xc, efc = layers.Codes(atoms, [28]), layers.Codes(bonds, [4])
    step = CountLayerStep(plan, layer, [3, 3, 3, 3], clamp=True)
    ids, y, status = step(node_ptr, edge_ptr, ei, xc, efc, mn, me)
    step.check_status()
    # the composition of the separate entry points
    ep = packs.new_edge_pack(E, dev)
    ids_ref, _, idc = count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev,
                                  encode=([3, 3, 3, 3], True), encoded_pack=(ep, 0), encoded_rows=False)
    with torch.no_grad():
        y_ref = layer(layers.Codes(atoms, [28]), ei, identifiers=idc, degrees=torch.zeros(N, device=dev), edge_features=layers.Codes(bonds, [4]))
    assert torch.equal(ids, ids_ref)
    assert torch.equal(y, y_ref)
    # a second step into caller-owned outputs, after a weight update: the prepared weights follow the parameters
    with torch.no_grad():
        layer.update_fn.fc[0].weight.mul_(1.5)
    ids2, y2 = torch.empty_like(ids), torch.empty_like(y)
    step(node_ptr, edge_ptr, ei, xc, efc, mn, me, ids_out=ids2, out=y2)
    with torch.no_grad():
        y_ref2 = layer(layers.Codes(atoms, [28]), ei, identifiers=idc, degrees=torch.zeros(N, device=dev), edge_features=layers.Codes(bonds, [4]))
    assert torch.equal(ids2, ids_ref) and torch.equal(y2, y_ref2) and not torch.equal(y2, y_ref)


def test_registered_side_outputs_feed_the_plain_layer_call():
    """count_batch_side(register=True), then the ordinary layer call: the layer finds the CSR and the packs (one kernel launch, no index
    build, no encoder launch) and gives the rows of the unregistered composition."""
    from gsn_amd import layers, flags
    from gsn_amd._index import _CSR_CACHE
    from gsn_amd.counting import CountPlan, count_batch_side
    b, node_ptr, edge_ptr, ei, atoms, bonds = _zinc(300, 21)
    N = b.num_nodes
    mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
    dev = _dev()
    plan = CountPlan.get(_cycles(), "edge", False)
    torch.manual_seed(1)
    layer = layers.GSN_edge_sparse(flow="source_to_target", **CTOR).to(dev).eval()
    deg = torch.zeros(N, device=dev)
    xc, efc = layers.Codes(atoms, [28]), layers.Codes(bonds, [4])
    layer._folded_first_weight(28)                                      # (made once per weight version: two small products)
    _CSR_CACHE.clear()
    r = count_batch_side(plan, node_ptr, edge_ptr, ei, mn, me, id_classes=[3, 3, 3, 3], x_codes=xc, ef_codes=efc, csr_row=layer._sel())
    flags.KERNEL_TIMER = {}
    try:
        with torch.no_grad():
            y = layer(xc, ei, identifiers=r["id_codes"], degrees=deg, edge_features=efc)
        torch.cuda.synchronize()
        launched = sorted(flags.KERNEL_TIMER.keys())
    finally:
        flags.KERNEL_TIMER = None
    assert launched == ["layer_fused"], launched
    _CSR_CACHE.clear()
    with torch.no_grad():
        y_ref = layer(layers.Codes(atoms, [28]), ei, identifiers=layers.Codes(r["ids"].clone(), [3, 3, 3, 3], clamp=True), degrees=deg,
                      edge_features=layers.Codes(bonds, [4]))
    assert torch.equal(y, y_ref)


@no_cache
def test_step_is_capturable_and_replays_on_refilled_inputs():
    from gsn_amd import layers
    from gsn_amd.counting import CountPlan
    from gsn_amd.step import CountLayerStep
    b, node_ptr, edge_ptr, ei, atoms, bonds = _zinc(128, 77)
    mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
    dev = _dev()
    plan = CountPlan.get(_cycles(), "edge", False)
    torch.manual_seed(2)
    layer = layers.GSN_edge_sparse(flow="source_to_target", **CTOR).to(dev).eval()
    xc, efc = layers.Codes(atoms, [28]), layers.Codes(bonds, [4])
    step = CountLayerStep(plan, layer, [3, 3, 3, 3])
    ids0, y0, _ = step(node_ptr, edge_ptr, ei, xc, efc, mn, me)
    ids_g, y_g = torch.empty_like(ids0), torch.empty_like(y0)
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        step(node_ptr, edge_ptr, ei, xc, efc, mn, me, ids_out=ids_g, out=y_g)
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step(node_ptr, edge_ptr, ei, xc, efc, mn, me, ids_out=ids_g, out=y_g)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(ids_g, ids0) and torch.equal(y_g, y0)
    # other codes in the same buffers: the replay encodes them again (nothing input-keyed is cached)
    atoms.copy_((atoms + 3) % 28); bonds.copy_((bonds + 1) % 4)
    g.replay()
    ids1, y1, _ = step(node_ptr, edge_ptr, ei, xc, efc, mn, me)
    torch.cuda.synchronize()
    assert torch.equal(ids_g, ids1) and torch.equal(y_g, y1) and not torch.equal(y1, y0)


def test_a_partition_that_under_declares_its_graphs_is_refused():
    """kernels that size their tiles by the declared max_nodes trusted it silently; checked where that is free (host pointers) or asked for."""
    from gsn_amd import layers
    b, node_ptr, edge_ptr, ei, atoms, bonds = _zinc(12, 4)
    mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
    assert layers.set_graph_partition(ei, node_ptr, edge_ptr, mn, me, check=True)
    with pytest.raises(ValueError, match="larger than the declared max"):
        layers.set_graph_partition(ei, node_ptr, edge_ptr, mn - 1, me, check=True)
    with pytest.raises(ValueError, match="larger than the declared max"):
        layers.set_graph_partition(ei, torch.from_numpy(b.node_ptr), torch.from_numpy(b.edge_ptr), mn, me - 1, check=False)      # host pointers: always
    assert layers.set_graph_partition(ei, node_ptr, edge_ptr, mn - 1, me, check=False)      # device pointers, no check asked for: the caller vouches
    layers.set_graph_partition(ei, node_ptr, edge_ptr, mn, me, check=False)


@no_cache
def test_graphed_step_on_persistent_codes_encodes_them_again_in_every_replay():
    """Codes objects that survive across forwards were tagged with their packs during the warm-up; the capture found valid tags,
    recorded only the layer kernel, and a replay after ``codes.copy_(new)`` read the warm-up's rows.  Tags now carry a capture epoch."""
    from gsn_amd import layers
    from gsn_amd.counting import CountPlan, count_batch
    from gsn_amd.graphs import GraphedStep
    b, node_ptr, edge_ptr, ei, atoms, bonds = _zinc(96, 31)
    N = b.num_nodes
    dev = _dev()
    plan = CountPlan.get(_cycles(), "edge", False)
    ids, _ = count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, device=dev)
    torch.manual_seed(4)
    layer = layers.GSN_edge_sparse(flow="source_to_target", **CTOR).to(dev).eval()
    deg = torch.zeros(N, device=dev)
    xc, efc, idc = layers.Codes(atoms, [28]), layers.Codes(bonds, [4]), layers.Codes(ids, [3, 3, 3, 3], clamp=True)

    def fn():
        with torch.no_grad():
            return layer(xc, ei, identifiers=idc, degrees=deg, edge_features=efc)
    gs = GraphedStep(fn, warmup=2)
    y0 = gs().clone()
    atoms.copy_((atoms + 5) % 28); bonds.copy_((bonds + 1) % 4); ids.copy_((ids + 1) % 3)
    y1 = gs().clone()
    with torch.no_grad():
        y_ref = layer(layers.Codes(atoms.clone(), [28]), ei, identifiers=layers.Codes(ids.clone(), [3, 3, 3, 3], clamp=True), degrees=deg,
                      edge_features=layers.Codes(bonds.clone(), [4]))
    assert torch.equal(y1, y_ref) and not torch.equal(y0, y1)

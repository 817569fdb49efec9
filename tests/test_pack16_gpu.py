"""The one-launch `general` layer on exact fp16 row packs (gsn_amd.packs, gsn_layer_fused_fwd_pack16_hip, csrc/layer_rp.hip) on a real
MI355X: against the oracle's fp32 restatement of the reference layers (GSN_edge_sparse.py:82-170, GSN_sparse.py:93-176,
MPNN_edge_sparse.py:110-151), against the fp32 kernel of the same package, on the tile shapes of test_fused_gpu.py; the tags that
select the kernel (current / stale / claimed by another tensor / refused for inexact rows); the counting kernel as a producer."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CTOR = dict(d_in=28, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=128,
            d_up=128, d_h=[128], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target")


def _elementwise_ok(got, ref, rtol=1e-5):
    floor = rtol * ref.abs().amax(dim=1, keepdim=True)
    return bool(((got - ref).abs() <= rtol * ref.abs() + floor).all())


def _randomise_bn(layer, seed):
    g = torch.Generator().manual_seed(seed)
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.rand(m.running_mean.shape, generator=g) * 0.6 - 0.3)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.rand(m.bias.shape, generator=g) * 0.6 - 0.3)


def _traced(fn):
    """(result, stderr trace of the launches)"""
    import tempfile
    os.environ["GSN_CHAIN_TRACE"] = "1"
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        saved = os.dup(2)
        os.dup2(tmp.fileno(), 2)
        try:
            out = fn()
            torch.cuda.synchronize()
        finally:
            os.dup2(saved, 2)
            os.close(saved)
            os.environ.pop("GSN_CHAIN_TRACE", None)
        tmp.seek(0)
        return out, tmp.read().decode(errors="replace")


def _run(cls, ctor, x, ei, ids, ef, seed=0, pack=True):
    """-> (packed-row kernel output, fp32 kernel output, oracle, trace of the packed call)"""
    from gsn_amd import layers, packs
    from oracle import oracle
    torch.manual_seed(seed)
    layer = getattr(layers, cls)(**ctor)
    _randomise_bn(layer, seed + 1)
    layer.eval()
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    kw = dict(identifiers=ids, degrees=None)
    if ef is not None:
        kw["edge_features"] = ef
    ref = oracle.layer_forward(cls, ctor, sd, x, ei, training=False, **kw)
    layer.cuda()
    xg = x.cuda()
    kwg = dict(identifiers=None if ids is None else ids.cuda(), degrees=torch.zeros(x.shape[0], device="cuda"))
    if ef is not None:
        kwg["edge_features"] = ef.cuda()
    eig = ei.cuda()
    with torch.no_grad():
        layers._CSR_CACHE.clear()
        y32 = layer(xg, eig, **kwg)                      # untagged: the fp32 kernel
        if pack:
            packs.node_pack(xg)
            per_edge = [t for t in (kwg.get("identifiers") if "d_id" in ctor else None, kwg.get("edge_features")) if t is not None]
            if per_edge:
                packs.edge_pack(per_edge)
        y16, trace = _traced(lambda: layer(xg, eig, **kwg))
    return y16.cpu(), y32.cpu(), ref, trace


def _zinc(n_graphs, seed):
    from gsn_amd import synth
    b = synth.zinc_shape_batch(n_graphs, seed=seed)
    x = torch.nn.functional.one_hot(torch.from_numpy(b.atom_type), 28).float()
    ef = torch.nn.functional.one_hot(torch.from_numpy(b.bond_type), 4).float()
    return b, x, ef, torch.from_numpy(b.edge_index)


@pytest.mark.parametrize("n_graphs", [1, 3, 64, 4096])
def test_packed_layer_zinc_shape(n_graphs):
    """layer 0 of BASELINE config 2 (GSN_edge_sparse general / local, d = 128) on tagged one-hot inputs"""
    b, x, ef, ei = _zinc(n_graphs, seed=120 + n_graphs)
    ids = (torch.rand(b.num_edges, 12, generator=torch.Generator().manual_seed(1)) < 0.2).float()
    y16, y32, ref, trace = _run("GSN_edge_sparse", CTOR, x, ei, ids, ef, seed=3)
    assert "layer_fused_kernel_rp" in trace, trace[-400:]
    assert _elementwise_ok(y16, ref), float((y16 - ref).abs().max() / ref.abs().max())
    assert _elementwise_ok(y16, y32, rtol=2e-6)


def test_untagged_inputs_take_the_fp32_kernel():
    b, x, ef, ei = _zinc(16, seed=7)
    ids = (torch.rand(b.num_edges, 12, generator=torch.Generator().manual_seed(1)) < 0.2).float()
    y16, y32, ref, trace = _run("GSN_edge_sparse", CTOR, x, ei, ids, ef, seed=3, pack=False)
    assert "layer_fused_kernel_rr" in trace and "layer_fused_kernel_rp" not in trace, trace[-400:]
    assert torch.equal(y16, y32)


@pytest.mark.parametrize("cls,d_in,d_id,d_ef", [("GSN_edge_sparse", 28, 12, 4), ("GSN_edge_sparse", 8, 8, 8), ("GSN_edge_sparse", 4, 15, 1),
                                               ("GSN_edge_sparse", 20, 3, 5), ("GSN_sparse", 28, 16, 0), ("GSN_sparse", 12, 5, 0),
                                               ("MPNN_edge_sparse", 28, 0, 4), ("MPNN_edge_sparse", 16, 0, 16), ("MPNN_sparse", 24, 0, 0)])
def test_packed_layer_classes_and_widths(cls, d_in, d_id, d_ef):
    """the four layer classes (with / without identifiers, with / without edge features), node rows of 4 .. 28 columns (the in-degree
    slot moves through the chunks and lane halves), edge-level rows of 0 .. 16 columns; values 0, 1, -1, 0.5, 1.5, 2^-10 (all exact)"""
    from gsn_amd import synth
    b = synth.zinc_shape_batch(200, seed=31 + d_in)
    g = torch.Generator().manual_seed(d_in * 100 + d_id)
    vals = torch.tensor([0.0, 0.0, 0.0, 1.0, -1.0, 0.5, 1.5, 2.0 ** -10])
    x = vals[torch.randint(0, len(vals), (b.num_nodes, d_in), generator=g)]
    ids = vals[torch.randint(0, len(vals), (b.num_edges, d_id), generator=g)] if d_id else None
    ef = vals[torch.randint(0, len(vals), (b.num_edges, d_ef), generator=g)] if d_ef else None
    ctor = dict(d_in=d_in, d_degree=1, degree_as_tag=False, retain_features=True, d_msg=128, d_up=128, d_h=[128], seed=0,
                activation_name="relu", bn=True, msg_kind="general", flow="source_to_target")
    if d_id:
        ctor.update(d_id=d_id, id_scope="local")
    if d_ef:
        ctor.update(d_ef=d_ef)
    y16, y32, ref, trace = _run(cls, ctor, x, torch.from_numpy(b.edge_index), ids, ef, seed=d_in + d_ef)
    assert "layer_fused_kernel_rp" in trace, trace[-400:]
    assert _elementwise_ok(y16, ref), float((y16 - ref).abs().max() / ref.abs().max())
    assert _elementwise_ok(y16, y32, rtol=2e-6)


def test_packed_layer_dense_hub_isolated():
    """tiles with several blocks (dense graphs), a hub whose in-degree (3 000) is not an fp16 value, runs of isolated nodes, edge-less
    graphs at the start / end, duplicate edges; both flows; all-zero node rows (the row scale then comes from S / the degree alone)"""
    from gsn_amd import synth
    rng = np.random.default_rng(3)
    graphs = [(5, np.zeros((2, 0), dtype=np.int64)), synth.er_graph(40, 300, 1)]
    star = np.stack([np.zeros(3000, dtype=np.int64), np.arange(1, 3001)])
    graphs.append((3100, np.concatenate([star, star[::-1]], axis=1)))
    graphs.append(synth.zinc_shape_graph(rng))
    graphs.append((3, np.array([[0, 1, 0, 1, 2, 1], [1, 0, 1, 0, 1, 2]], dtype=np.int64)))
    graphs.append(synth.er_graph(128, 1000, 2))
    graphs.append((70, np.zeros((2, 0), dtype=np.int64)))
    b = synth.collate(graphs)
    g = torch.Generator().manual_seed(9)
    N, E = b.num_nodes, b.num_edges
    x = torch.nn.functional.one_hot(torch.randint(0, 28, (N,), generator=g), 28).float()
    x[torch.rand(N, generator=g) < 0.2] = 0.0
    ef = torch.nn.functional.one_hot(torch.randint(0, 4, (E,), generator=g), 4).float()
    ids = torch.randint(0, 2, (E, 12), generator=g).float()
    ei = torch.from_numpy(b.edge_index)
    for flow in ("source_to_target", "target_to_source"):
        y16, y32, ref, trace = _run("GSN_edge_sparse", dict(CTOR, flow=flow), x, ei, ids, ef, seed=11)
        assert "layer_fused_kernel_rp" in trace, trace[-400:]
        assert _elementwise_ok(y16, ref), (flow, float((y16 - ref).abs().max() / ref.abs().max()))
        assert _elementwise_ok(y16, y32, rtol=2e-6)


def test_identity_activations_and_no_batchnorm():
    b, x, ef, ei = _zinc(300, seed=41)
    ids = (torch.rand(b.num_edges, 12, generator=torch.Generator().manual_seed(2)) < 0.3).float()
    for act, bn in (("identity", True), ("relu", False), ("identity", False)):
        y16, y32, ref, trace = _run("GSN_edge_sparse", dict(CTOR, activation_name=act, bn=bn), x, ei, ids, ef, seed=13)
        assert "layer_fused_kernel_rp" in trace, trace[-400:]
        assert _elementwise_ok(y16, ref), (act, bn, float((y16 - ref).abs().max() / ref.abs().max()))


def test_tags_follow_the_tensors():
    """a tensor written since it was packed, a pack whose columns another tensor has claimed, a released tag: the fp32 kernel, same result"""
    from gsn_amd import layers, packs
    b, x, ef, ei = _zinc(64, seed=51)
    ids = (torch.rand(b.num_edges, 12, generator=torch.Generator().manual_seed(3)) < 0.3).float()
    torch.manual_seed(0)
    layer = layers.GSN_edge_sparse(**CTOR).cuda().eval()
    xg, eg, ig, eig = x.cuda(), ef.cuda(), ids.cuda(), ei.cuda()
    deg = torch.zeros(b.num_nodes, device="cuda")
    call = lambda: layer(xg, eig, identifiers=ig, degrees=deg, edge_features=eg)
    with torch.no_grad():
        y0 = call()
        packs.node_pack(xg)
        ep = packs.edge_pack([ig, eg])
        y1, t1 = _traced(call)
        assert "layer_fused_kernel_rp" in t1
        assert _elementwise_ok(y1.cpu(), y0.cpu(), rtol=2e-6)
        # (a) in-place write: the version counter moves, the pack no longer describes the tensor
        ig[0, 0] = 1.0 - ig[0, 0]
        y2, t2 = _traced(call)
        assert "layer_fused_kernel_rp" not in t2 and "layer_fused_kernel_rr" in t2
        ref2 = layer(xg, eig, identifiers=ig.clone(), degrees=deg, edge_features=eg.clone())
        assert torch.equal(y2, ref2)
        # (b) re-packed: current again
        packs.edge_pack([ig, eg], pack=ep)
        y3, t3 = _traced(call)
        assert "layer_fused_kernel_rp" in t3 and _elementwise_ok(y3.cpu(), ref2.cpu(), rtol=2e-6)
        # (c) another tensor claims the identifier columns of the same pack
        other = (torch.rand(b.num_edges, 12, device="cuda") < 0.5).float()
        packs.edge_pack([other], pack=ep)
        y4, t4 = _traced(call)
        assert "layer_fused_kernel_rp" not in t4 and torch.equal(y4, ref2)
        # (d) the layer called with THAT tensor reads the pack again (columns 0..11 are its, 12..15 still the edge features')
        y5, t5 = _traced(lambda: layer(xg, eig, identifiers=other, degrees=deg, edge_features=eg))
        assert "layer_fused_kernel_rp" in t5
        ref5 = layer(xg.clone(), eig, identifiers=other.clone(), degrees=deg, edge_features=eg.clone())
        assert _elementwise_ok(y5.cpu(), ref5.cpu(), rtol=2e-6)
        # (e) released by hand
        packs.release(xg)
        y6, t6 = _traced(lambda: layer(xg, eig, identifiers=other, degrees=deg, edge_features=eg))
        assert "layer_fused_kernel_rp" not in t6 and torch.equal(y6, ref5)


def test_inexact_rows_are_refused():
    from gsn_amd import packs
    x = torch.rand(100, 28, device="cuda")                       # not fp16 values
    with pytest.raises(ValueError):
        packs.node_pack(x)
    assert getattr(x, "_gsn_pack16", None) is None
    x2 = torch.full((10, 8), 2.0, device="cuda")                 # exact, but not below 2
    with pytest.raises(ValueError):
        packs.node_pack(x2)
    x3 = torch.full((10, 8), float("inf"), device="cuda")
    with pytest.raises(ValueError):
        packs.edge_pack([x3])


def test_pack_contents():
    """gsn_pack16_rows_hip: values, zero padding, the constant-1 column, column ranges of a shared pack"""
    from gsn_amd import packs
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-1, 2, (333, 20), generator=g).float().cuda()
    p = packs.node_pack(x)
    assert p.shape == (333, 32) and p.dtype == torch.float16
    assert torch.equal(p[:, :20].float(), x) and bool((p[:, 20:31] == 0).all()) and bool((p[:, 31] == 1).all())
    a = torch.randint(0, 2, (77, 5), generator=g).float().cuda()
    c = torch.randint(0, 2, (77, 9), generator=g).float().cuda()
    ep = packs.edge_pack([a, c])
    assert torch.equal(ep[:, :5].float(), a) and torch.equal(ep[:, 5:14].float(), c) and bool((ep[:, 14:] == 0).all())
    assert packs.lookup(x, []) is not None and packs.lookup(x, [a, c])[1] is ep
    assert packs.lookup(x, [c, a]) is None and packs.lookup(x, [a]) is not None


@pytest.mark.parametrize("n_graphs", [1, 300, 5000])
def test_counting_kernel_writes_the_pack(n_graphs):
    """gsn_count_encode_pack16_hip: the encoded identifier rows as fp32 AND as fp16 in columns 0..11 of an edge pack, bit-identical values;
    the layer fed those tensors runs the packed-row kernel and agrees with the oracle"""
    import networkx as nx
    from gsn_amd import layers, packs, synth
    from gsn_amd.counting import CountPlan, count_batch
    from oracle import oracle
    b = synth.zinc_shape_batch(n_graphs, seed=77 + n_graphs)
    pats = [list(nx.cycle_graph(k).edges) for k in range(3, 7)]
    plan = CountPlan.get(pats, "edge", False)
    dev = torch.device("cuda")
    E = b.num_edges
    ep = packs.new_edge_pack(E, dev)
    ef = torch.nn.functional.one_hot(torch.from_numpy(b.bond_type), 4).float().to(dev)
    # the layer concatenates identifiers first: identifiers in columns 0..11 (written by the counting kernel), edge features in 12..15
    ids64, _, enc = count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index, ids_are_global=True, device=dev, encode=([3, 3, 3, 3], True),
                                encoded_pack=(ep, 0))
    packs._pack_rows(ef, ep, 12, -1, True)
    packs.claim(ef, ep, 12)
    assert torch.equal(enc, torch.nn.functional.one_hot(ids64.clamp(max=2), 3).reshape(E, 12).float())
    assert torch.equal(ep[:, :12].float(), enc) and torch.equal(ep[:, 12:].float(), ef)
    x = torch.nn.functional.one_hot(torch.from_numpy(b.atom_type), 28).float().to(dev)
    packs.node_pack(x)
    torch.manual_seed(1)
    layer = layers.GSN_edge_sparse(**CTOR).eval()
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    ei = torch.from_numpy(b.edge_index)
    ref = oracle.layer_forward("GSN_edge_sparse", CTOR, sd, x.cpu(), ei, identifiers=enc.cpu(), degrees=None, edge_features=ef.cpu(), training=False)
    layer.cuda()
    with torch.no_grad():
        y, trace = _traced(lambda: layer(x, ei.to(dev), identifiers=enc, degrees=torch.zeros(b.num_nodes, device=dev), edge_features=ef))
    assert "layer_fused_kernel_rp" in trace, trace[-400:]
    assert _elementwise_ok(y.cpu(), ref), float((y.cpu() - ref).abs().max() / ref.abs().max())
    # a second call WITHOUT the pack into the same buffer drops the tag
    count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index, ids_are_global=True, device=dev, encode=([3, 3, 3, 3], True), encoded_out=enc, counts=False)
    assert packs.lookup(x, [enc, ef]) is None


@pytest.mark.parametrize("n_graphs,ids_kind", [(1, "codes"), (200, "codes"), (200, "tagged"), (4096, "tagged")])
def test_integer_codes_go_straight_into_packs(n_graphs, ids_kind):
    """Codes inputs (the one-hot encoder's INPUT, utils_graph_learning.py:170-187): gsn_one_hot_pack16_hip writes their encodings into the
    packs (bit-identical to the fp16 of the dense one-hot), no fp32 one-hot of x / the bond types is made, the packed-row kernel runs, and
    the result equals the oracle on the dense encodings.  Identifiers as Codes over the int64 counts, or as the counting kernel's tagged rows."""
    import networkx as nx
    from gsn_amd import layers, packs, synth
    from gsn_amd.counting import CountPlan, count_batch
    from oracle import oracle
    b = synth.zinc_shape_batch(n_graphs, seed=300 + n_graphs)
    dev = torch.device("cuda")
    E, N = b.num_edges, b.num_nodes
    plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", False)
    xc = layers.Codes(torch.from_numpy(b.atom_type).to(dev), [28])
    efc = layers.Codes(torch.from_numpy(b.bond_type).to(dev), [4])
    x = torch.nn.functional.one_hot(torch.from_numpy(b.atom_type), 28).float()
    ef = torch.nn.functional.one_hot(torch.from_numpy(b.bond_type), 4).float()
    if ids_kind == "codes":
        ids64, _ = count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index, ids_are_global=True, device=dev)
        ids_in = layers.Codes(ids64, [3, 3, 3, 3], clamp=True)
        enc = torch.nn.functional.one_hot(ids64.clamp(max=2), 3).reshape(E, 12).float()
    else:
        ep = packs.new_edge_pack(E, dev)
        _, _, enc = count_batch(plan, b.node_ptr, b.edge_ptr, b.edge_index, ids_are_global=True, device=dev, encode=([3, 3, 3, 3], True),
                                counts=False, encoded_pack=(ep, 0))
        ids_in = enc
    torch.manual_seed(4)
    layer = layers.GSN_edge_sparse(**CTOR)
    _randomise_bn(layer, 5)
    layer.eval()
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    ei = torch.from_numpy(b.edge_index)
    ref = oracle.layer_forward("GSN_edge_sparse", CTOR, sd, x, ei, identifiers=enc.cpu(), degrees=None, edge_features=ef, training=False)
    layer.cuda()
    with torch.no_grad():
        y, trace = _traced(lambda: layer(xc, ei.to(dev), identifiers=ids_in, degrees=torch.zeros(N, device=dev), edge_features=efc))
    assert "layer_fused_kernel_rp" in trace, trace[-400:]
    assert _elementwise_ok(y.cpu(), ref), float((y.cpu() - ref).abs().max() / ref.abs().max())
    npk, epk = xc._pack16[0], efc._pack16[0]
    assert torch.equal(npk[:, :28].float().cpu(), x) and bool((npk[:, 28:31] == 0).all()) and bool((npk[:, 31] == 1).all())
    assert torch.equal(epk[:, 12:].float().cpu(), ef) and torch.equal(epk[:, :12].float(), enc)
    assert xc._dense is None and efc._dense is None          # (no dense one-hot was made)


def test_one_hot_pack_out_of_range_and_odd_boundaries():
    """gsn_one_hot_pack16_hip: a code outside its classes leaves its segment zero (clamp: nearest class), as gsn_one_hot_hip; segments at
    odd columns and of odd width keep their neighbours."""
    from gsn_amd import layers, packs
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    codes = torch.randint(-2, 7, (1000, 3), generator=g).to(dev)
    for clamp in (False, True):
        cd = layers.Codes(codes, [5, 3, 4], clamp=clamp)
        pack = torch.full((1000, packs.EDGE_COLS), 7.0, dtype=torch.float16, device=dev)
        if not clamp:
            with pytest.raises(IndexError):          # (the reference's F.one_hot raises on such a code)
                packs._pack_codes(cd, pack, 3, -1)
        packs._pack_codes(cd, pack, 3, -1, check=False)
        dense = layers.one_hot_identifiers(codes, [5, 3, 4], clamp=clamp)
        assert torch.equal(pack[:, 3:15].float(), dense)
        assert bool((pack[:, :3] == 7).all()) and bool((pack[:, 15:] == 7).all())


def test_codes_rewritten_in_place_are_encoded_again():
    """A Codes object keeps the pack made from it; writing its code tensor in place (a reused input buffer) moves the version counter and the next
    layer call encodes the new codes instead of reading the pack of the old ones."""
    from gsn_amd import layers, packs
    from oracle import oracle
    b, x, ef, ei = _zinc(50, seed=7)
    dev = torch.device("cuda")
    atom = torch.from_numpy(b.atom_type).to(dev)
    xc = layers.Codes(atom, [28])
    efc = layers.Codes(torch.from_numpy(b.bond_type).to(dev), [4])
    ids = (torch.rand(b.num_edges, 12, generator=torch.Generator().manual_seed(1)) < 0.2).float()
    torch.manual_seed(6)
    layer = layers.GSN_edge_sparse(**CTOR)
    _randomise_bn(layer, 2)
    layer.eval()
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    layer.cuda()
    idg = ids.to(dev)
    packs.edge_pack([idg])
    kw = dict(identifiers=idg, degrees=torch.zeros(b.num_nodes, device=dev), edge_features=efc)
    with torch.no_grad():
        layer(xc, ei.to(dev), **kw)
        first = xc._pack16[0]
        xc.codes.copy_((xc.codes + 5) % 28)                    # the same buffer, other atoms
        y = layer(xc, ei.to(dev), **kw)
    assert xc._pack16[0] is not first
    x2 = torch.nn.functional.one_hot(xc.codes[:, 0].cpu(), 28).float()
    ref = oracle.layer_forward("GSN_edge_sparse", CTOR, sd, x2, ei, identifiers=ids, degrees=None, edge_features=ef, training=False)
    assert _elementwise_ok(y.cpu(), ref), float((y.cpu() - ref).abs().max() / ref.abs().max())

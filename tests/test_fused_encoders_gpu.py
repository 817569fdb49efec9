"""GNN_OGB with GSN_edge_sparse_ogb layers: the identifier and edge-feature embeddings of a layer summed by ONE launch over the concatenated code
columns (models._fused_edge_encoding, flags.FUSE_EDGE_ENCODERS) against the two-encoder path the reference takes
(models_graph_classification_ogb_original.py:213-223, GSN_edge_sparse_ogb.py:103-106: relu(x_j + id_e + e_e)): same prediction, same
gradients up to the reassociation of the fp32 sum, one per-edge stream in the propagate launch, one embedding launch per layer."""
import importlib.util
import os
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(fuse, d, batch, train):
    from gsn_amd import flags
    dev = torch.device("cuda", 0)
    m = _script("train_step_molhiv")
    old = flags.FUSE_EDGE_ENCODERS
    flags.FUSE_EDGE_ENCODERS = fuse
    try:
        torch.manual_seed(7)
        model, data, params, opt, loss_of, N, E = m.build(types.SimpleNamespace(batch=batch, layers=3, d=d, optimizer="sgd"), dev, 0, dropout=0.0)
        model.train(train)
        flags.KERNEL_TIMER = {}
        out = model(data)
        launches = {k: len(v) for k, v in flags.KERNEL_TIMER.items()}
        flags.KERNEL_TIMER = None
        loss = (out * torch.linspace(-1.0, 1.0, out.numel(), device=dev).view_as(out)).sum()
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        return out.detach().clone(), grads, launches
    finally:
        flags.FUSE_EDGE_ENCODERS = old
        flags.KERNEL_TIMER = None


@pytest.mark.parametrize("d,batch", [(64, 24), (300, 48), (136, 200)])
@pytest.mark.parametrize("train", [True, False])
def test_fused_edge_encoders_match_the_two_encoder_path(d, batch, train):
    o0, g0, l0 = _run(False, d, batch, train)
    o1, g1, l1 = _run(True, d, batch, train)
    scale = o0.abs().max().item() + 1e-30
    assert (o1 - o0).abs().max().item() <= 2e-5 * scale
    assert set(g0) == set(g1)
    gscale = max(g.abs().max().item() for g in g0.values())
    for n in g0:
        s = g0[n].abs().max().item()
        if train:
            # train mode: relu(BatchNorm(.)) puts ~1e6 pre-activations of a step around zero, a handful of them within the reassociation noise
            # of the message sum (1e-7): each such ReLU switches a whole gradient path (scripts/gpu/diag_fused_enc.py: ONE unit of the virtual
            # node's MLP moved every gradient of the step by 2e-3).  The element-wise bar belongs to the kink-free tests below; here: the same
            # gradient field up to such events -- a wrong table, a dropped term or a transposed index is an O(1) difference
            num = (g1[n] - g0[n]).norm().item()
            assert num <= 3e-2 * g0[n].norm().item() + 1e-5 * gscale * g0[n].numel() ** 0.5, (n, num, g0[n].norm().item())
        else:
            # eval mode: BatchNorm on its running statistics is a fixed affine map, the VN rows are not normalised over 24..200 graphs -- sharp
            tol = 2e-5 * s + 1e-6 * gscale
            assert (g1[n] - g0[n]).abs().max().item() <= tol, (n, (g1[n] - g0[n]).abs().max().item(), s)
    # one embedding launch per layer instead of two (3 layers: the atom encoder + the virtual node's + 3 instead of + 6)
    assert l1["embed_fwd"] == l0["embed_fwd"] - 3, (l0, l1)


def _encoders(d, dev):
    from gsn_amd.encoding import DiscreteEmbedding
    kw = {"seed": 0, "activation_mlp": "relu", "bn_mlp": True, "aggr": "sum", "features_scope": "full"}
    torch.manual_seed(3)
    id_enc = DiscreteEmbedding("embedding", 4, [3, 5, 4, 7], d, **kw).to(dev)
    ef_enc = DiscreteEmbedding("bond_encoder", 3, None, d, **kw).to(dev)
    return id_enc, ef_enc


@pytest.mark.parametrize("d,E", [(300, 5000), (64, 37), (132, 0)])
def test_fused_encoding_is_the_sum_of_the_two_encoders_and_so_are_its_table_gradients(d, E):
    """The op is linear in the tables: no kink, sharp bars (forward: two fp32 roundings of association; gradients: sums of the same rows)."""
    from gsn_amd import encoding, models
    dev = torch.device("cuda", 0)
    id_enc, ef_enc = _encoders(d, dev)
    g = torch.Generator(device="cpu").manual_seed(E + d)
    ids = torch.stack([torch.randint(0, n, (E,), generator=g) for n in (3, 5, 4, 7)], 1).to(dev)
    ef = torch.stack([torch.randint(0, n, (E,), generator=g) for n in encoding.BOND_FEATURE_DIMS], 1).to(dev)
    up = torch.randn(E, d, generator=g).to(dev)
    ref = id_enc(ids) + ef_enc(ef)
    (ref * up).sum().backward()
    want = [p.grad.clone() for p in list(id_enc.parameters()) + list(ef_enc.parameters())]
    for p in list(id_enc.parameters()) + list(ef_enc.parameters()):
        p.grad = None
    got = models._fused_edge_encoding({}, id_enc, ef_enc, ids, ef)
    assert got is not None and got.shape == ref.shape
    if E:
        assert (got - ref).abs().max().item() <= 4e-7 * ref.abs().max().item()
    (got * up).sum().backward()
    have = [p.grad for p in list(id_enc.parameters()) + list(ef_enc.parameters())]
    for w, h in zip(want, have):
        assert h is not None and h.shape == w.shape
        assert (h - w).abs().max().item() <= 2e-6 * (w.abs().max().item() + 1e-30) + 1e-30
    # not taken: float identifiers (already encoded), a concatenating encoder, different widths
    assert models._fused_edge_encoding({}, id_enc, ef_enc, ids.float(), ef) is None
    other = encoding.DiscreteEmbedding("embedding", 4, [3, 5, 4, 7], d, seed=0, activation_mlp="relu", bn_mlp=True, aggr="concat", features_scope="full").to(dev)
    assert models._fused_edge_encoding({}, other, ef_enc, ids, ef) is None
    narrow = encoding.DiscreteEmbedding("bond_encoder", 3, None, d + 4, seed=0, activation_mlp="relu", bn_mlp=True, aggr="sum", features_scope="full").to(dev)
    assert models._fused_edge_encoding({}, id_enc, narrow, ids, ef) is None


@pytest.mark.parametrize("d", [300, 136, 64])
@pytest.mark.parametrize("train", [True, False])
def test_ogb_layer_on_one_summed_edge_stream_equals_the_two_stream_layer(d, train):
    """relu(x_j + (id_e + e_e)) against relu((x_j + id_e) + e_e) on inputs whose sums are EXACT in fp32 (multiples of 1/64 below 8): the two
    associations give the same bits, so every ReLU decides alike and the layer's output and input gradients agree to the atomics' noise --
    the two-stream launch of relu_sum3_kernel / the generic kernel against the three-stream one, forward and adjoint."""
    from gsn_amd import layers, synth
    dev = torch.device("cuda", 0)
    b = synth.zinc_shape_batch(300, seed=11)
    N, E = b.num_nodes, b.num_edges
    ei = torch.from_numpy(b.edge_index).to(dev)
    g = torch.Generator(device="cpu").manual_seed(d)
    dy = lambda *shape: (torch.randint(-128, 129, shape, generator=g).float() / 64.0).to(dev)
    x, ids, ef = dy(N, d).requires_grad_(), dy(E, d), dy(E, d)
    lay = layers.GSN_edge_sparse_ogb(d_in=d, d_ef=d, d_id=d, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=d, d_up=d,
                                     d_h=[2 * d], seed=0, activation_name="elu", bn=True, train_eps=True, flow="source_to_target").to(dev).train(train)
    up = torch.randn(N, d, generator=g).to(dev)
    outs = []
    for kw in ({"identifiers": ids.clone().requires_grad_(), "edge_features": ef.clone().requires_grad_()},
               {"identifiers": (ids + ef).requires_grad_(), "edge_features": None}):
        x.grad = None
        lay.zero_grad(set_to_none=True)
        y = lay(x, ei, degrees=None, **kw)
        (y * up).sum().backward()
        outs.append((y.detach().clone(), x.grad.clone(), kw["identifiers"].grad.clone(),
                     {n: p.grad.clone() for n, p in lay.named_parameters() if p.grad is not None}))
    (y0, gx0, gi0, gp0), (y1, gx1, gi1, gp1) = outs
    assert (y1 - y0).abs().max().item() <= 2e-6 * y0.abs().max().item()
    assert (gx1 - gx0).abs().max().item() <= 2e-5 * gx0.abs().max().item()
    assert (gi1 - gi0).abs().max().item() <= 2e-5 * gi0.abs().max().item()          # (the masked per-edge gradient: one tensor for both encoders)
    gs = max(v.abs().max().item() for v in gp0.values())
    for n in gp0:
        assert (gp1[n] - gp0[n]).abs().max().item() <= 2e-5 * gp0[n].abs().max().item() + 1e-6 * gs, n


def test_layer_rejects_missing_edge_features_without_identifiers():
    from gsn_amd import layers
    dev = torch.device("cuda", 0)
    lay = layers.MPNN_edge_sparse_ogb(d_in=8, d_ef=8, d_degree=1, degree_as_tag=False, retain_features=True, d_msg=8, d_up=8, d_h=[16], seed=0,
                                      activation_name="relu", bn=False, flow="source_to_target").to(dev)
    x = torch.randn(5, 8, device=dev)
    ei = torch.tensor([[0, 1, 2, 3], [1, 2, 3, 4]], device=dev)
    with pytest.raises(RuntimeError, match="edge_features missing"):
        lay(x, ei, degrees=None, identifiers=None, edge_features=None)

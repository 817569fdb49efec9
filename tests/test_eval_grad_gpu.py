"""Round 3, VERDICT r02 item 8: gradients in eval mode and the gin / ogb layers' own term on the HIP kernels.

(a) A gradient asked for with the layer in eval mode (BatchNorm on its running statistics: frozen-BN fine-tuning, saliency maps) runs on
    the same adjoint kernels as training -- gsn_bn_act_bwd_hip (train_bn = 0 / 2), gsn_wgrad_hip, gsn_propagate_pad_bwd_hip -- and
    never on the PyTorch twin; checked against autograd over the oracle's plain-PyTorch restatement of the reference layers
    (graph_filters/*.py with models_misc.py:41-45 under model.eval()).
(b) (1 + eps) * self + sum of messages and the central encoders' concatenations (GSN_sparse.py:157-163, GSN_edge_sparse.py:95-109,
    GSN_edge_sparse_ogb.py:103-106, utils_graph_learning.py:232-260) happen inside the propagate kernel: the forward of a gin / ogb
    layer launches no ATen kernel (torch profiler), and equals the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BASE = dict(d_degree=1, degree_as_tag=False, retain_features=True, seed=0, activation_name="elu", bn=True, flow="source_to_target")
# (class, constructor arguments, identifiers per edge?, has edge features?)
CASES = {
    "general_local": ("GSN_edge_sparse", dict(d_in=8, d_ef=3, d_id=5, id_scope="local", d_msg=16, d_up=16, d_h=[16], msg_kind="general"), True, True),
    "general_global": ("GSN_sparse", dict(d_in=8, d_id=5, id_scope="global", d_msg=16, d_up=24, d_h=[16], msg_kind="general"), False, False),
    "general_no_hidden": ("GSN_sparse", dict(d_in=8, d_id=5, id_scope="local", d_msg=16, d_up=16, d_h=[], msg_kind="general"), True, False),
    "mpnn_edge_general": ("MPNN_edge_sparse", dict(d_in=8, d_ef=4, d_msg=16, d_up=16, d_h=[16], msg_kind="general"), None, True),
    "gin_global": ("GSN_sparse", dict(d_in=8, d_id=5, id_scope="global", d_msg=None, d_up=16, d_h=[16], msg_kind="gin", train_eps=True, eps=0.25,
                                      id_embedding="one_hot_encoder", extend_dims=True), False, False),
    "gin_local_one_hot_extend": ("GSN_edge_sparse", dict(d_in=8, d_ef=3, d_id=5, id_scope="local", d_msg=None, d_up=16, d_h=[16], msg_kind="gin",
                                                         train_eps=True, eps=0.5, id_embedding="one_hot_encoder", edge_embedding="one_hot_encoder",
                                                         extend_dims=True), True, True),
    "gin_local_embedding_extend": ("GSN_edge_sparse", dict(d_in=8, d_ef=4, d_id=4, id_scope="local", d_msg=None, d_up=16, d_h=[16], msg_kind="gin",
                                                           train_eps=True, eps=0.125, id_embedding="embedding", edge_embedding="embedding",
                                                           extend_dims=True), True, True),
    "gin_local_plain": ("GSN_edge_sparse", dict(d_in=8, d_ef=4, d_id=4, id_scope="local", d_msg=None, d_up=16, d_h=[16], msg_kind="gin",
                                                train_eps=False, eps=0.0, id_embedding="one_hot_encoder", edge_embedding="embedding",
                                                extend_dims=False), True, True),
    "mpnn_edge_gin": ("MPNN_edge_sparse", dict(d_in=8, d_ef=3, d_msg=None, d_up=16, d_h=[16], msg_kind="gin", train_eps=True, eps=0.1,
                                               edge_embedding="one_hot_encoder", extend_dims=True), None, True),
    "ogb_local": ("GSN_edge_sparse_ogb", dict(d_in=8, d_ef=8, d_id=8, id_scope="local", d_msg=None, d_up=8, d_h=[16], msg_kind="ogb", train_eps=True,
                                              eps=0.3), True, True),
    "ogb_global": ("GSN_edge_sparse_ogb", dict(d_in=8, d_ef=8, d_id=8, id_scope="global", d_msg=None, d_up=8, d_h=[16], msg_kind="ogb",
                                               train_eps=True, eps=0.3), False, True),
    "mpnn_ogb": ("MPNN_edge_sparse_ogb", dict(d_in=8, d_ef=8, d_msg=None, d_up=8, d_h=[16], msg_kind="ogb", train_eps=False, eps=0.0), None, True),
}


def _setup(name, graphs=6, seed=3):
    from gsn_amd import layers, synth
    cls, extra, ids_per_edge, has_ef = CASES[name]
    ctor = dict(BASE, **extra)
    b = synth.zinc_shape_batch(graphs, seed=seed)
    n, E = b.num_nodes, b.num_edges
    torch.manual_seed(11)
    layer = getattr(layers, cls)(**ctor)
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.3, 0.3)
    x = torch.randn(n, ctor["d_in"])
    kw = {}
    if ids_per_edge is not None:
        kw["identifiers"] = torch.randn(E if ids_per_edge else n, ctor["d_id"])
    if has_ef:
        kw["edge_features"] = torch.randn(E, ctor["d_ef"])
    return cls, ctor, layer, x, torch.from_numpy(b.edge_index), kw


def _oracle_grads(cls, ctor, layer, x, ei, kw, w=None):
    from oracle import oracle
    pn = {k for k, _ in layer.named_parameters()}
    sd = {k: v.clone().requires_grad_(k in pn) for k, v in layer.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    kwr = {k: v.clone().requires_grad_(True) for k, v in kw.items()}
    yr = oracle.layer_forward(cls, ctor, sd, xr, ei, degrees=None, training=False, **kwr)
    if w is None:
        w = torch.randn_like(yr)
    (yr * w).sum().backward()
    return yr.detach(), w, xr.grad, {k: v.grad for k, v in kwr.items()}, {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}


@pytest.mark.parametrize("name", sorted(CASES))
def test_eval_mode_gradients_on_the_hip_adjoints(name):
    from gsn_amd import layers
    cls, ctor, layer, x, ei, kw = _setup(name)
    layer.eval()
    yr, w, gx_ref, gkw_ref, gw_ref = _oracle_grads(cls, ctor, layer, x, ei, kw)
    layer.cuda()
    twin_calls, native_calls = [], []
    orig_t, orig_n = layers._HipWithTorchBackward.apply, layers._HipWithNativeBackward.apply
    layers._HipWithTorchBackward.apply = staticmethod(lambda *a, **k: (twin_calls.append(1), orig_t(*a, **k))[1])
    layers._HipWithNativeBackward.apply = staticmethod(lambda *a, **k: (native_calls.append(1), orig_n(*a, **k))[1])
    try:
        xg = x.cuda().requires_grad_(True)
        kwg = {k: v.cuda().requires_grad_(True) for k, v in kw.items()}
        extra = {} if "identifiers" in kwg else {"identifiers": None}       # (MPNN_edge_sparse_ogb.py reads the key as the reference does)
        y = layer(xg, ei.cuda(), degrees=torch.zeros(x.shape[0], device="cuda"), **kwg, **extra)
        (y * w.cuda()).sum().backward()
    finally:
        layers._HipWithTorchBackward.apply, layers._HipWithNativeBackward.apply = orig_t, orig_n
    assert not twin_calls, "an eval-mode gradient went through the PyTorch twin"
    assert native_calls, "the eval-mode forward did not take the fused forward + native recompute route"
    ymax = float(yr.abs().max())
    assert float((y.detach().cpu() - yr).abs().max()) <= 1e-5 * ymax
    assert float((xg.grad.cpu() - gx_ref).abs().max()) <= 2e-5 * float(gx_ref.abs().max())
    for k, g in gkw_ref.items():
        assert float((kwg[k].grad.cpu() - g).abs().max()) <= 2e-5 * max(float(g.abs().max()), 1e-3 * float(gx_ref.abs().max())), k
    gmax = max(float(g.abs().max()) for g in gw_ref.values())
    got = dict(layer.named_parameters())
    assert set(gw_ref) <= set(got)
    for k, g in gw_ref.items():
        assert got[k].grad is not None, k
        assert float((got[k].grad.cpu() - g).abs().max()) <= 2e-5 * gmax, (k, float((got[k].grad.cpu() - g).abs().max()), gmax)
    # the running statistics are untouched by an eval-mode backward (its recompute must not count as a training step)
    for k, v in layer.state_dict().items():
        if "running_" in k or "num_batches" in k:
            ref = dict(_setup(name)[2].state_dict())[k]
            assert torch.equal(v.cpu(), ref), k


@pytest.mark.parametrize("name", sorted(k for k in CASES if "gin" in k or "ogb" in k))
def test_gin_and_ogb_forward_launch_no_aten_kernel(name):
    """The own term, eps, the central encoders' rows and zero columns are arguments of gsn_propagate_self_fwd_hip: an eval forward of a gin /
    ogb layer consists of our launches only (the torch profiler sees no device kernel at all -- ours do not go through ATen)."""
    from oracle import oracle
    cls, ctor, layer, x, ei, kw = _setup(name, graphs=40, seed=5)
    layer.eval()
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    ref = oracle.layer_forward(cls, ctor, sd, x, ei, degrees=None, training=False, **kw)
    layer.cuda()
    xg, eig = x.cuda(), ei.cuda()
    kwg = {k: v.cuda() for k, v in kw.items()}
    deg = torch.zeros(x.shape[0], device="cuda")
    kwg.setdefault("identifiers", None)
    with torch.no_grad():
        y = layer(xg, eig, degrees=deg, **kwg)          # (first call: CSR build, derived-weight caches, constant rows)
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
            y = layer(xg, eig, degrees=deg, **kwg)
            torch.cuda.synchronize()
    assert float((y.cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    aten = [e.key for e in prof.key_averages() if e.key.startswith("aten::") and e.key not in
            ("aten::empty", "aten::empty_strided", "aten::empty_like", "aten::zeros", "aten::zero_", "aten::fill_", "aten::view", "aten::reshape",
             "aten::slice", "aten::select", "aten::as_strided", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::detach", "aten::alias",
             "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::t", "aten::transpose", "aten::_unsafe_view", "aten::resize_",
             "aten::lift_fresh", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense")]
    compute = [k for k in aten if k in ("aten::cat", "aten::mul", "aten::add", "aten::add_", "aten::mul_", "aten::copy_", "aten::index_select",
                                        "aten::zeros_like", "aten::clone", "aten::sub", "aten::rsub")]
    assert not compute, "ATen compute ops in a %s forward: %s" % (name, compute)


def test_propagate_self_term_and_padding_vs_torch():
    """gsn_propagate_self_fwd_hip / gsn_propagate_pad_bwd_hip on their own: widths that take the float4 and the scalar path, every
    combination of per-node / single-row self blocks, forward and gradients against plain tensor ops."""
    from gsn_amd import layers, synth
    b = synth.zinc_shape_batch(50, seed=9)
    n, E = b.num_nodes, b.num_edges
    ei = torch.from_numpy(b.edge_index).cuda()
    sel = 1
    src, tgt = ei[0], ei[1]
    torch.manual_seed(2)
    for da, db, dc, pads, single in [(8, 4, 4, (0, 0), (False, True, True)), (7, 5, 3, (1, 1), (False, True, True)),
                                     (16, 8, 0, (0, 0), (False, False)), (12, 0, 4, (0, 0), (False, True)), (5, 6, 0, (1, 0), (False, True))]:
        a = torch.randn(n, da, device="cuda", requires_grad=True)
        bb = torch.randn(E, db, device="cuda", requires_grad=True) if db else None
        c = torch.randn(E, dc, device="cuda", requires_grad=True) if dc else None
        widths = [da] + ([pads[0] + db] if db else []) + ([pads[1] + dc] if dc else [])
        selfs = [torch.randn(1 if s else n, w, device="cuda", requires_grad=True) for w, s in zip(widths, single)]
        eps = torch.tensor([0.37], device="cuda", requires_grad=True)
        out = layers.propagate(0, ei, sel, n, a=a, b=bb, c=c, selfs=selfs, eps=eps, pads=pads)
        parts = [a[src]]
        if db:
            parts += [torch.zeros(E, pads[0], device="cuda"), bb]
        if dc:
            parts += [torch.zeros(E, pads[1], device="cuda"), c]
        msgs = torch.cat(parts, -1)
        ref = (1 + eps) * torch.cat([s.expand(n, -1) for s in selfs], -1) + torch.zeros(n, msgs.shape[1], device="cuda").index_add_(0, tgt, msgs)
        assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
        w = torch.randn_like(ref)
        ins = [t for t in [a, bb, c, eps] + selfs if t is not None]
        g_got = torch.autograd.grad((out * w).sum(), ins)
        g_ref = torch.autograd.grad((ref * w).sum(), ins)
        for g, r in zip(g_got, g_ref):
            assert g.shape == r.shape
            assert torch.allclose(g, r, rtol=1e-4, atol=1e-4 * max(1.0, float(r.abs().max()))), (da, db, dc, float((g - r).abs().max()))
    # relu-sum kind (ogb): self blocks are ADDED
    d = 12
    a = torch.randn(n, d, device="cuda", requires_grad=True)
    bn_ = torch.randn(n, d, device="cuda", requires_grad=True)
    c = torch.randn(E, d, device="cuda", requires_grad=True)
    eps = torch.tensor([0.2], device="cuda", requires_grad=True)
    out = layers.propagate(1, ei, sel, n, a=a, b=bn_, c=c, b_per_node=True, selfs=[a, bn_], eps=eps)
    ref = (1 + eps) * (a + bn_) + torch.zeros(n, d, device="cuda").index_add_(0, tgt, torch.relu(a[src] + bn_[src] + c))
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
    w = torch.randn_like(ref)
    for g, r in zip(torch.autograd.grad((out * w).sum(), [a, bn_, c, eps]), torch.autograd.grad((ref * w).sum(), [a, bn_, c, eps])):
        assert torch.allclose(g, r, rtol=1e-4, atol=1e-4 * max(1.0, float(r.abs().max())))


@pytest.mark.parametrize("m_rows,n_out,widths", [(5000, 600, (300,)), (777, 70, (33, 5, 128)), (33, 300, (600,)), (100000, 128, (260,)), (17, 9, (7,))])
def test_weight_gradient_bf16x6_vs_fp64(m_rows, n_out, widths):
    """gsn_wgrad_hip (wgrad_bf16_kernel: six exact bf16 plane products, contraction over the rows) against an fp64 product: the error of
    an fp32 FMA loop, at magnitudes from 1e-6 to 1e6 per column, ragged row counts, concatenated blocks, tiles past the matrix edges."""
    import ctypes
    from gsn_amd import _abi
    torch.manual_seed(m_rows)
    gh = torch.randn(m_rows, n_out, device="cuda") * torch.logspace(-6, 6, n_out, device="cuda")
    blocks = [torch.randn(m_rows, w, device="cuda") * torch.logspace(-3, 3, w, device="cuda") for w in widths]
    k_total = sum(widths)
    gw = torch.zeros(n_out, k_total, device="cuda")
    arr = (_abi.gsn_block * len(blocks))()
    for i, t in enumerate(blocks):
        arr[i].data = t.data_ptr(); arr[i].idx = None; arr[i].idx32 = None; arr[i].width = t.shape[1]
    _abi.check(_abi.lib().gsn_wgrad_hip(m_rows, n_out, gh.data_ptr(), len(blocks), arr, gw.data_ptr(), _abi.current_stream()), "gsn_wgrad_hip")
    x = torch.cat(blocks, 1)
    ref = gh.double().t() @ x.double()
    # element-wise against the product of the column magnitudes (what an fp32 accumulation of m_rows terms can promise)
    bound = (gh.double().abs().t() @ x.double().abs())
    err = ((gw.double() - ref).abs() / bound.clamp_min(1e-300)).max().item()
    ref32 = (gh.t() @ x).double()
    err32 = ((ref32 - ref).abs() / bound.clamp_min(1e-300)).max().item()
    assert err <= max(2.0 * err32, 3e-7), (err, err32)


def test_add_by_graph_vs_torch():
    """gsn_add_gathered_hip: x + table[batch] and its adjoint (identity / sum readout), float4 and scalar widths; an index outside the
    table gives a NaN row."""
    from gsn_amd import layers
    torch.manual_seed(4)
    for n, g, d in [(1000, 37, 300), (513, 5, 7), (64, 64, 128)]:
        batch = torch.sort(torch.randint(0, g, (n,), device="cuda")).values
        x = torch.randn(n, d, device="cuda", requires_grad=True)
        tab = torch.randn(g, d, device="cuda", requires_grad=True)
        y = layers.add_by_graph(x, tab, batch)
        ref = x + tab[batch]
        assert torch.equal(y, ref)
        w = torch.randn_like(ref)
        gx, gt = torch.autograd.grad((y * w).sum(), [x, tab])
        rx, rt = torch.autograd.grad((ref * w).sum(), [x, tab])
        assert torch.equal(gx, rx)
        assert torch.allclose(gt, rt, rtol=1e-5, atol=1e-5)
    bad = torch.tensor([0, 1, 5], device="cuda")
    y = layers.add_by_graph(torch.zeros(3, 4, device="cuda"), torch.ones(2, 4, device="cuda"), bad)
    assert torch.isnan(y[2]).all() and torch.equal(y[:2], torch.ones(2, 4, device="cuda"))


@pytest.mark.parametrize("seed", [116, 179, 207, 254, 264, 7022, 7403])
def test_edge_less_batches_give_the_zero_gradients_pytorch_gives(seed):
    """Regression (found by tests/soak_grads.py): a batch without a single edge -- every class / kind, train and eval mode.  The forward
    equals the oracle, and every parameter and input that only acts through the edges gets a ZERO gradient (not None, not an error):
    zero-row stages launch nothing, the per-edge blocks are empty tensors without a pointer."""
    import soak_grads
    r = soak_grads.one_case(seed)
    assert r is not None
    desc, bad = r
    assert " E=0" in desc, desc
    assert not bad, (desc, bad)


@pytest.mark.parametrize("name", ["general_local", "gin_local_embedding_extend", "ogb_local"])
def test_eval_mode_gradients_with_frozen_batchnorm_parameters(name):
    """Frozen-BatchNorm fine-tuning: the layer in eval mode AND gamma / beta without gradients -- the eval-mode stage is then a per-column
    affine epilogue of the product (no pre-BatchNorm rows are kept: _DenseStagesFn's `affine` stage, gsn_bn_act_bwd_hip with train_bn = 0
    and coef = gamma * invstd); every other gradient equals autograd over the oracle, BatchNorm's parameters get none."""
    cls, ctor, layer, x, ei, kw = _setup(name)
    layer.eval()
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.requires_grad_(False); m.bias.requires_grad_(False)
    yr, w, gx_ref, gkw_ref, gw_ref = _oracle_grads(cls, ctor, layer, x, ei, kw)
    gw_ref = {k: g for k, g in gw_ref.items() if ".bn." not in k}          # (the oracle differentiates everything: BatchNorm's are not asked for here)
    layer.cuda()
    xg = x.cuda().requires_grad_(True)
    kwg = {k: v.cuda().requires_grad_(True) for k, v in kw.items()}
    y = layer(xg, ei.cuda(), degrees=torch.zeros(x.shape[0], device="cuda"), **kwg)
    (y * w.cuda()).sum().backward()
    assert float((y.detach().cpu() - yr).abs().max()) <= 1e-5 * float(yr.abs().max())
    assert float((xg.grad.cpu() - gx_ref).abs().max()) <= 2e-5 * float(gx_ref.abs().max())
    gmax = max(float(g.abs().max()) for g in gw_ref.values())
    got = dict(layer.named_parameters())
    for k, g in gw_ref.items():
        assert got[k].grad is not None, k
        assert float((got[k].grad.cpu() - g).abs().max()) <= 2e-5 * gmax, k
    for k, p in got.items():
        if ".bn." in k:
            assert p.grad is None, k

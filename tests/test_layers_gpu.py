"""HP-2 parity on a real MI355X: HIP kernels (through the C ABI) against the golden vectors produced by the reference
layers and against plain PyTorch fp32.  Tolerance (BASELINE north_star: "within 1e-5 relative"): forward outputs are checked
ELEMENT-WISE, |got - ref| <= 1e-5 |ref| + 1e-5 max|ref row| (`elementwise_ok`: relative, with an absolute floor of 1e-5 of
the row's largest magnitude for elements that are small through cancellation), next to the max-normalised figure `rel_err`."""
import numpy as np
import pytest
import torch

from helpers import case_names, layer_case

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel_err(a, b, floor=1e-30):
    return (a - b).abs().max().item() / max(b.abs().max().item(), floor)


def elementwise_ok(got, ref, rtol=TOL):
    """|got - ref| <= rtol |ref| + rtol * max|ref row|, every element"""
    got, ref = got.reshape(ref.shape[0], -1), ref.reshape(ref.shape[0], -1)
    return bool(((got - ref).abs() <= rtol * ref.abs() + rtol * ref.abs().amax(dim=1, keepdim=True)).all())


def test_csr_and_propagate_vs_torch():
    from gsn_amd.layers import build_csr, propagate
    torch.manual_seed(0)
    dev = "cuda"
    for N, E in ((1, 0), (7, 3), (300, 2000), (5000, 40000), (70000, 150000)):
        ei = torch.randint(0, N, (2, E), device=dev)
        seg, perm = build_csr(ei[1], N)
        deg = torch.bincount(ei[1], minlength=N)
        assert torch.equal(seg[1:] - seg[:-1], deg.int())
        if E:
            assert torch.equal(ei[1][perm.long()], torch.sort(ei[1], stable=True)[0])
            assert torch.equal(perm.long(), torch.sort(ei[1], stable=True)[1])   # stable: original order inside a segment
        for (da, db, dc, per_node) in ((28, 12, 4, False), (7, 5, 3, True), (0, 128, 0, False), (13, 0, 4, False), (300, 0, 0, False)):
            a = torch.randn(N, da, device=dev) if da else None
            b = (torch.randn(N if per_node else E, db, device=dev) if db else None)
            c = torch.randn(E, dc, device=dev) if dc else None
            out = propagate(0, ei, 1, N, a=a, b=b, c=c, b_per_node=per_node)
            parts = []
            if da: parts.append(a[ei[0]])
            if db: parts.append(b[ei[0]] if per_node else b)
            if dc: parts.append(c)
            ref = torch.zeros(N, da + db + dc, device=dev).index_add_(0, ei[1], torch.cat(parts, 1)) if E else torch.zeros(N, da + db + dc, device=dev)
            assert rel_err(out, ref) < TOL if E else (out == 0).all()
        for d, per_node in ((8, False), (300, True), (33, False)):
            a = torch.randn(N, d, device=dev); b = torch.randn(N if per_node else E, d, device=dev); c = torch.randn(E, d, device=dev)
            out = propagate(1, ei, 0, N, a=a, b=b, c=c, b_per_node=per_node)   # target_to_source: aggregate at row 0, gather at row 1
            msg = torch.relu(a[ei[1]] + (b[ei[1]] if per_node else b) + c)
            ref = torch.zeros(N, d, device=dev).index_add_(0, ei[0], msg)
            if E:
                assert rel_err(out, ref) < TOL


def test_propagate_backward_vs_autograd():
    from gsn_amd.layers import propagate
    torch.manual_seed(1)
    dev = "cuda"
    N, E = 500, 3000
    ei = torch.randint(0, N, (2, E), device=dev)
    for kind, (da, db, dc), per_node in ((0, (6, 5, 3), False), (0, (6, 5, 0), True), (1, (9, 9, 9), False), (1, (9, 9, 9), True), (0, (0, 16, 0), False)):
        a = torch.randn(N, da, device=dev, requires_grad=True) if da else None
        b = torch.randn(N if per_node else E, db, device=dev, requires_grad=True) if db else None
        c = torch.randn(E, dc, device=dev, requires_grad=True) if dc else None
        out = propagate(kind, ei, 1, N, a=a, b=b, c=c, b_per_node=per_node)
        w = torch.randn_like(out)
        (out * w).sum().backward()
        got = [t.grad.clone() if t is not None else None for t in (a, b, c)]
        for t in (a, b, c):
            if t is not None: t.grad = None
        parts = [a[ei[0]] if a is not None else None, (b[ei[0]] if per_node else b) if b is not None else None, c]
        parts = [p for p in parts if p is not None]
        msg = torch.cat(parts, 1) if kind == 0 else torch.relu(sum(parts))
        ref = torch.zeros(N, msg.shape[1], device=dev).index_add(0, ei[1], msg)
        (ref * w).sum().backward()
        for g, t in zip(got, (a, b, c)):
            if t is not None:
                assert rel_err(g, t.grad) < TOL


def test_linear_kernel_vs_torch():
    """Asymmetric data (transposes would show), ragged sizes, gathered + concatenated inputs, every epilogue."""
    from gsn_amd.layers import _linear_hip
    torch.manual_seed(2)
    dev = "cuda"
    for M, widths, n_out in ((1, [3], 5), (130, [28, 28, 12, 4], 128), (1000, [13, 1], 70), (257, [64, 33, 7, 2, 40], 300), (4096, [156], 128)):
        R = 50
        blocks, cols = [], []
        for i, w in enumerate(widths):
            if i % 2 == 0 and M > 1:
                data = torch.randn(R, w, device=dev) * (1 + i)
                idx = torch.randint(0, R, (M,), device=dev)
                blocks.append((data, idx)); cols.append(data[idx])
            else:
                data = torch.randn(M, w, device=dev) + 0.5
                blocks.append((data, None)); cols.append(data)
        X = torch.cat(cols, 1).double()
        K = X.shape[1]
        W = torch.randn(n_out, K, device=dev) / K ** 0.5
        bias = torch.randn(n_out, device=dev)
        h = X @ W.double().T + bias.double()
        y = _linear_hip(blocks, W, bias, None, None, None, 0, M)
        assert rel_err(y.double(), h) < TOL
        mean = torch.randn(n_out, device=dev); scale = torch.rand(n_out, device=dev) + 0.5; shift = torch.randn(n_out, device=dev)
        z = (h - mean.double()) * scale.double() + shift.double()
        for act, fn in ((1, torch.relu), (2, torch.nn.functional.elu), (3, torch.tanh), (0, lambda t: t)):
            y = _linear_hip(blocks, W, bias, mean, scale, shift, act, M)
            assert rel_err(y.double(), fn(z)) < TOL
        stats = torch.zeros(2, n_out, dtype=torch.float64, device=dev)
        _linear_hip(blocks, W, bias, None, None, None, 0, M, out=False, stats=stats)
        # the kernel sums fp32 h values (each carrying ~1e-7 relative rounding) in fp64
        assert torch.allclose(stats[0], h.sum(0), rtol=1e-5, atol=1e-5 * M)
        assert torch.allclose(stats[1], (h * h).sum(0), rtol=1e-5, atol=1e-5 * M)


def _build(c):
    from gsn_amd import flags, layers
    layer = getattr(layers, c["cls"])(**c["ctor"])
    layer.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in c.items() if k.startswith("sd/")})
    return layer.cuda().train(c["train"])


def _inputs(c, grad=False):
    t = lambda k: torch.from_numpy(c[k]).cuda() if k in c else None
    x, ids, ef = t("x"), t("identifiers"), t("edge_features")
    if grad:
        for v in (x, ids, ef):
            if v is not None:
                v.requires_grad_(True)
    kw = {"degrees": t("degrees")}
    if ids is not None or c["cls"] == "MPNN_edge_sparse_ogb":
        kw["identifiers"] = ids
    if ef is not None:
        kw["edge_features"] = ef
    return x, t("edge_index"), kw, ids, ef


@pytest.mark.parametrize("name", case_names("layers"))
def test_layer_forward_golden(name):
    c = layer_case(name)
    layer = _build(c)
    x, ei, kw, _, _ = _inputs(c)
    with torch.no_grad():
        y = layer(x, ei, **kw)
    ref = torch.from_numpy(c["y"]).cuda()
    assert y.shape == ref.shape and y.is_cuda
    assert rel_err(y, ref) < TOL, rel_err(y, ref)
    assert elementwise_ok(y, ref), name
    if c["train"]:
        sd = layer.state_dict()
        for k in c:
            if k.startswith("sd_after/") and "num_batches" not in k:
                assert torch.allclose(sd[k[9:]].cpu(), torch.from_numpy(c[k]), rtol=1e-4, atol=1e-6), k
            elif k.startswith("sd_after/"):
                assert int(sd[k[9:]]) == int(c[k])


@pytest.mark.parametrize("name", case_names("layers"))
def test_layer_backward_golden(name):
    """input and parameter gradients of EVERY reference-generated case -- the molecule graphs and the hub / isolated-node graph --
    against the reference's own autograd gradients (g_x, g_identifiers, g_edge_features, gp/* of layers.npz)"""
    c = layer_case(name)
    layer = _build(c)
    x, ei, kw, ids, ef = _inputs(c, grad=True)
    y = layer(x, ei, **kw)
    assert rel_err(y.detach(), torch.from_numpy(c["y"]).cuda()) < TOL
    (y * torch.from_numpy(c["w"]).cuda()).sum().backward()
    BT = 2e-5  # gradients go through train-mode BN statistics twice; a little looser than the forward bar (1e-5)
    # a bias in front of a train-mode BatchNorm has an exactly-zero gradient: both sides hold only rounding noise there,
    # whose size follows the layer's overall gradient scale -> floor the denominator at 2% of the largest parameter gradient
    FL = 0.02 * max([float(np.abs(c[k]).max()) for k in c if k.startswith("gp/")] + [0.5])
    assert rel_err(x.grad, torch.from_numpy(c["g_x"]).cuda().reshape(x.shape), FL) < BT
    if ids is not None:
        assert rel_err(ids.grad, torch.from_numpy(c["g_identifiers"]).cuda(), FL) < BT
    if ef is not None:
        assert rel_err(ef.grad, torch.from_numpy(c["g_edge_features"]).cuda(), FL) < BT
    for k, p in layer.named_parameters():
        if "gp/" + k in c:
            assert p.grad is not None, k
            assert rel_err(p.grad, torch.from_numpy(c["gp/" + k]).cuda(), FL) < BT, k


def test_layer_vs_oracle_big_batch():
    """A 4096-graph ZINC-shaped batch at the real layer-0 widths (BASELINE config 2) against the oracle's fp32 torch restatement."""
    from gsn_amd import flags, layers, synth
    from oracle import oracle
    torch.manual_seed(5)
    b = synth.zinc_shape_batch(4096, seed=11)
    N, E = b.num_nodes, b.num_edges
    ctor = dict(d_in=28, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=128,
                d_up=128, d_h=[128], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target")
    layer = layers.GSN_edge_sparse(**ctor)
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.3, 0.3)
    layer.eval()
    x = torch.nn.functional.one_hot(torch.from_numpy(b.atom_type), 28).float()
    ef = torch.nn.functional.one_hot(torch.from_numpy(b.bond_type), 4).float()
    ids = (torch.rand(E, 12) < 0.2).float()
    ei = torch.from_numpy(b.edge_index)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    ref = oracle.layer_forward("GSN_edge_sparse", ctor, sd, x, ei, identifiers=ids, degrees=None, edge_features=ef, training=False)
    layer.cuda()
    with torch.no_grad():
        y = layer(x.cuda(), ei.cuda(), identifiers=ids.cuda(), degrees=torch.zeros(N, device="cuda"), edge_features=ef.cuda())
    assert rel_err(y.cpu(), ref) < TOL
    assert elementwise_ok(y.cpu(), ref)


@pytest.mark.parametrize("cls,kind", [("GSN_edge_sparse", "general"), ("GSN_sparse", "gin"), ("MPNN_edge_sparse", "general")])
def test_narrow_layers_on_a_big_batch_vs_oracle(cls, kind):
    """d = 64 (BASELINE config 3's width) on 4096 graphs: 95 k vertices / 195 k edge rows, i.e. every persistent workgroup of the
    dense kernels handles several row tiles, with stages that have fewer output columns than a workgroup has column waves
    (regression: the waves without output columns of mlp_chain_kernel skipped their share of the next tile's loads -- rows
    past the first 128 x gridDim were computed from the previous tile's inputs; the small-batch goldens never got there)."""
    from gsn_amd import flags, layers, synth
    from oracle import oracle
    torch.manual_seed(8)
    b = synth.zinc_shape_batch(4096, seed=12)
    N, E = b.num_nodes, b.num_edges
    ctor = dict(d_in=28, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=64,
                d_up=64, d_h=[64], seed=0, activation_name="relu", bn=True, flow="source_to_target")
    if cls == "GSN_sparse":
        ctor.update(id_scope="global", msg_kind="gin", aggr="add", train_eps=False)
        ctor.pop("d_ef")
    else:
        ctor["msg_kind"] = kind
    layer = getattr(layers, cls)(**ctor)
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.3, 0.3)
    layer.eval()
    x = torch.nn.functional.one_hot(torch.from_numpy(b.atom_type), 28).float()
    ef = torch.nn.functional.one_hot(torch.from_numpy(b.bond_type), 4).float()
    ei = torch.from_numpy(b.edge_index)
    ids = (torch.rand(N if cls == "GSN_sparse" else E, 12) < 0.2).float()
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    kw = dict(identifiers=ids, degrees=None, training=False)
    if cls != "GSN_sparse":
        kw["edge_features"] = ef
    ref = oracle.layer_forward(cls, ctor, sd, x, ei, **kw)
    layer.cuda()
    with torch.no_grad():
        kw2 = dict(identifiers=ids.cuda(), degrees=torch.zeros(N, device="cuda"))
        if cls != "GSN_sparse":
            kw2["edge_features"] = ef.cuda()
        y = layer(x.cuda(), ei.cuda(), **kw2)
    assert rel_err(y.cpu(), ref) < TOL
    assert elementwise_ok(y.cpu(), ref)


def test_one_hot_identifiers_vs_torch():
    from gsn_amd.layers import one_hot_identifiers
    torch.manual_seed(3)
    v = torch.randint(0, 7, (5000, 4), device="cuda")
    ncls = [7, 9, 7, 8]
    got = one_hot_identifiers(v, ncls)
    ref = torch.cat([torch.nn.functional.one_hot(v[:, c], ncls[c]).float() for c in range(4)], 1)
    assert torch.equal(got, ref)
    got = one_hot_identifiers(v, [3, 3, 3, 3], clamp=True)
    ref = torch.nn.functional.one_hot(v.clamp(max=2), 3).reshape(5000, 12).float()
    assert torch.equal(got, ref)


def test_readout_pooling_vs_torch():
    """global_add/mean_pool_sparse (utils_graph_learning.py:23-41) on the segmented-sum kernel, forward and backward."""
    from gsn_amd.layers import global_add_pool_sparse, global_mean_pool_sparse
    torch.manual_seed(4)
    sizes = torch.randint(1, 40, (300,))
    batch = torch.repeat_interleave(torch.arange(300), sizes).cuda()
    x = torch.randn(batch.numel(), 64, device="cuda", requires_grad=True)
    ref = torch.zeros(300, 64, device="cuda").index_add(0, batch, x)
    got = global_add_pool_sparse(x, batch)
    assert rel_err(got, ref.detach()) < TOL
    w = torch.randn_like(got)
    (got * w).sum().backward()
    g1 = x.grad.clone(); x.grad = None
    (ref * w).sum().backward()
    assert rel_err(g1, x.grad) < TOL
    mean = global_mean_pool_sparse(x.detach(), batch)
    assert rel_err(mean, ref.detach() / sizes.cuda().float().unsqueeze(1)) < TOL


def test_readout_of_a_registered_sorted_batch():
    """layers.set_batch_partition: the readout of a collated batch takes its segment bounds from the node pointers (no index build
    -- the timer sees no csr_build launch); same sums, forward and backward, graphs without vertices included."""
    from gsn_amd import flags, layers
    torch.manual_seed(5)
    sizes = torch.randint(0, 40, (400,))
    sizes[0] = 0; sizes[-1] = 0; sizes[17] = 0
    batch = torch.repeat_interleave(torch.arange(400), sizes).cuda()
    node_ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(sizes, 0)]).cuda()
    x = torch.randn(batch.numel(), 96, device="cuda", requires_grad=True)
    ref = torch.zeros(400, 96, device="cuda").index_add(0, batch, x)
    layers.invalidate_caches()
    layers.set_batch_partition(batch, node_ptr)
    flags.KERNEL_TIMER = {}
    try:
        got = layers.global_add_pool_sparse(x, batch, 400)
        torch.cuda.synchronize()
        families = set(flags.KERNEL_TIMER)
    finally:
        flags.KERNEL_TIMER = None
    assert "csr_build" not in families and "propagate_fwd" in families
    assert rel_err(got, ref.detach()) < TOL
    w = torch.randn_like(got)
    (got * w).sum().backward()
    g1 = x.grad.clone(); x.grad = None
    (ref * w).sum().backward()
    assert rel_err(g1, x.grad) < TOL
    # an unregistered batch vector (here: a copy) takes the generic path and gives the same sums
    got2 = layers.global_add_pool_sparse(x.detach(), batch.clone(), 400)
    assert torch.equal(got2, got.detach())


def test_csr_cache_is_not_fooled_by_address_reuse():
    """Different graphs with identical shapes allocated at the same address must not share a cached CSR."""
    from gsn_amd.layers import propagate
    torch.manual_seed(6)
    N, E = 50, 300
    for trial in range(6):
        ei = torch.randint(0, N, (2, E), device="cuda")
        b = torch.randn(E, 16, device="cuda")
        out = propagate(0, ei, 1, N, b=b)
        ref = torch.zeros(N, 16, device="cuda").index_add_(0, ei[1], b)
        assert rel_err(out, ref) < TOL
        del ei, b, out, ref   # frees the blocks; the next iteration's tensors land on the same addresses
    ei = torch.randint(0, N, (2, E), device="cuda")
    b = torch.randn(E, 16, device="cuda")
    propagate(0, ei, 1, N, b=b)
    ei[1] = torch.randint(0, N, (E,), device="cuda")     # in-place change bumps the version counter
    out = propagate(0, ei, 1, N, b=b)
    assert rel_err(out, torch.zeros(N, 16, device="cuda").index_add_(0, ei[1], b)) < TOL


@pytest.mark.parametrize("shape", [
    # (M, block widths, hidden widths per stage, gathered?, bn, act)
    (1000, [72], [128], True, True, "relu"),            # single stage, K<=80 (8-wave kernel)
    (777, [100, 37], [64], False, True, "identity"),    # single stage, K in (80,160]
    (5000, [28, 128, 1], [128, 128], False, True, "relu"),   # two stages, K0=157 (the node chain of the bench layer)
    (333, [40], [100, 33], True, False, "relu"),        # two stages, K0<=80, odd widths, no bn
    (64, [16], [16, 8], False, True, "relu"),           # one exact tile
    (65, [3], [5], True, False, "identity"),            # tiny everything, tail row
    (4096, [160], [128, 96], False, True, "relu"),      # K0 at the limit
    (3000, [28, 128, 4], [128, 128], False, True, "relu"),   # bf16x6 node chain: float4 staging, NK0 = 10, NK1 = 8
    (2001, [32, 16], [64, 40], False, True, "relu"),    # bf16x6 node chain: K0 = 48 (NK0 = 5 planes padded), k1 = 64 (NK1 = 4)
    (1500, [28, 44], [100, 128], False, False, "identity"),  # bf16x6: K0 = 72, k1 = 100 (padded MID columns), no bn
    (999, [30, 50], [128, 128], False, True, "relu"),   # bf16x6 node chain, widths not multiples of 4: dword staging
    (31, [8], [16, 16], False, True, "relu"),           # less than one 32-row tile
])
def test_chain_kernel_shapes_vs_torch(shape):
    """gsn_mlp_chain_fwd_hip across its shape classes (kernel variants, tails, inactive waves) against fp64 torch."""
    from gsn_amd.layers import _Stage, _launch_stages, _chain_fits
    M, widths, hidden, gathered, bn, act = shape
    torch.manual_seed(sum(widths) + M)
    dev = "cuda"
    blocks, cols = [], []
    for i, w in enumerate(widths):
        if gathered and i % 2 == 0:
            data = torch.randn(97, w, device=dev); idx = torch.randint(0, 97, (M,), device=dev, dtype=torch.int32 if i % 4 == 0 else torch.int64)
            blocks.append((data, idx)); cols.append(data[idx.long()])
        else:
            data = torch.randn(M, w, device=dev); blocks.append((data, None)); cols.append(data)
    x = torch.cat(cols, 1).double()
    stages, k = [], x.shape[1]
    ref = x
    for s, n_out in enumerate(hidden):
        W = torch.randn(n_out, k, device=dev) / k ** 0.5
        b = torch.randn(n_out, device=dev)
        st = _Stage(W, b, None, act if s < len(hidden) - 1 or len(hidden) == 1 else "identity", blocks if s == 0 else ())
        h = ref @ W.double().T + b.double()
        if bn:
            mean, scale, shift = torch.randn(n_out, device=dev), torch.rand(n_out, device=dev) + 0.5, torch.randn(n_out, device=dev)
            st.bn_params = (mean, scale, shift)
            h = (h - mean.double()) * scale.double() + shift.double()
        ref = torch.relu(h) if st.act == "relu" else h
        stages.append(st); k = n_out
    assert _chain_fits(stages)
    y = _launch_stages(stages, M)
    assert y.shape == (M, hidden[-1])
    assert rel_err(y.double(), ref) < TOL
    # statistics pass of the last stage (pre-BN values)
    stats = torch.zeros(2, hidden[-1], dtype=torch.float64, device=dev)
    probe = stages[:-1] + [_Stage(stages[-1].weight, stages[-1].bias, None, "identity", stages[-1].blocks)]
    _launch_stages(probe, M, stats=stats)
    prev = x
    for st in stages[:-1]:
        h = prev @ st.weight.double().T + st.bias.double()
        if st.bn_params is not None:
            h = (h - st.bn_params[0].double()) * st.bn_params[1].double() + st.bn_params[2].double()
        prev = torch.relu(h) if st.act == "relu" else h
    hl = prev @ stages[-1].weight.double().T + stages[-1].bias.double()
    assert torch.allclose(stats[0], hl.sum(0), rtol=1e-5, atol=1e-5 * M)
    assert torch.allclose(stats[1], (hl * hl).sum(0), rtol=1e-5, atol=1e-5 * M)


@pytest.mark.parametrize("wx,we,n_out", [(20, 7, 96),      # K = 47: fp32 role-split kernel (blocks not float4-gatherable)
                                         (20, 8, 96),      # K = 48: bf16x6 kernel, NK16 = 3, partial second column half
                                         (28, 16, 128),    # K = 72: bf16x6, NK16 = 5, the predicate-free fast path
                                         (32, 0, 40),      # K = 64: bf16x6, NK16 = 4, fewer than 64 output columns
                                         (36, 8, 128)])    # K = 80: bf16x6 at its K limit
@pytest.mark.parametrize("N,E,hub", [(500, 4000, 0), (300, 5000, 700), (2000, 1500, 40), (64, 64, 64), (50, 17, 0)])
def test_fused_scatter_add_vs_torch(N, E, hub, wx, we, n_out):
    """The segmented-sum epilogue: empty segments, segments longer than one reduction range (atomics), tile tails -- on the
    fp32 and the bf16x6 edge-stage kernels."""
    from gsn_amd.layers import _Stage, run_stages, _csr_for
    torch.manual_seed(N + E)
    dev = "cuda"
    tgt = torch.randint(0, N, (E,), device=dev)
    if hub:
        tgt[:hub] = 3                                  # one target with a very long segment
    src = torch.randint(0, N, (E,), device=dev)
    ei = torch.stack([src, tgt], 0)
    x = torch.randn(N, wx, device=dev); ef = torch.randn(E, max(we, 1), device=dev)
    csr = _csr_for(ei, 1, N)
    K = 2 * wx + we
    W = torch.randn(n_out, K, device=dev) / K ** 0.5; b = torch.randn(n_out, device=dev)
    blocks = [(x, csr.tgt), (x, csr.src)] + ([(ef, csr.perm)] if we else [])
    st = _Stage(W, b, None, "relu", blocks)
    out = run_stages([st], E, False, csr=csr)
    assert out is not None and out.shape == (N, n_out)
    cat = [x[tgt], x[src]] + ([ef] if we else [])
    msg = torch.relu(torch.cat(cat, 1).double() @ W.double().T + b.double())
    ref = torch.zeros(N, n_out, dtype=torch.float64, device=dev).index_add_(0, tgt, msg)
    assert rel_err(out.double(), ref) < TOL
    assert (out[torch.bincount(tgt, minlength=N) == 0] == 0).all()


@pytest.mark.parametrize("act", ["relu", "elu", "tanh"])
def test_wide_mlp_train_mode_materialised_path(act):
    """Shapes outside the fused chain (n_out > 128) in train mode: pre-BN rows + statistics in one linear pass, then
    gsn_bn_act_hip -- against nn.Sequential of the same parameters, incl. the running statistics and a fused post-BN."""
    from gsn_amd import flags, layers
    torch.manual_seed(3)
    m = layers.mlp(70, 150, [200], 0, act, True).cuda().train()
    post = torch.nn.BatchNorm1d(150).cuda().train()
    ref = torch.nn.Sequential(torch.nn.Linear(70, 200), torch.nn.BatchNorm1d(200), layers.choose_activation(act) if act != "identity" else torch.nn.Identity(),
                              torch.nn.Linear(200, 150), torch.nn.BatchNorm1d(150), torch.nn.ReLU()).cuda().train()
    ref[0].load_state_dict(m.fc[0].state_dict()); ref[1].load_state_dict(m.bn[0].state_dict())
    ref[3].load_state_dict(m.fc[1].state_dict()); ref[4].load_state_dict(post.state_dict())
    x = torch.randn(777, 70, device="cuda")
    called = []
    from gsn_amd import _dense          # (run_stages looks the function up in its own module)
    orig = _dense._run_stages_materialised
    _dense._run_stages_materialised = lambda *a, **k: (called.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            y = m(x, post=(post, "relu"))
            want = ref(x)
    finally:
        _dense._run_stages_materialised = orig
    assert called, "wide train-mode stages must take the materialised path"
    assert float((y - want).abs().max()) <= 2e-5 * max(float(want.abs().max()), 1.0)
    for a, b in ((m.bn[0], ref[1]), (post, ref[4])):
        assert torch.allclose(a.running_mean, b.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(a.running_var, b.running_var, rtol=1e-4, atol=1e-6)
        assert int(a.num_batches_tracked) == 1


@pytest.mark.parametrize("shape", [(333, 70, [200], 150), (1000, 16, [24], 8), (257, 130, [64, 96], 33), (64, 5, [], 7)])
@pytest.mark.parametrize("act,bn,post", [("relu", True, True), ("elu", True, False), ("tanh", False, True), ("relu", False, False)])
def test_mlp_native_backward_matches_autograd(shape, act, bn, post):
    """gsn_bn_act_bwd_hip + gsn_wgrad_hip + the forward kernel on W^T against PyTorch autograd on the same parameters
    (train mode: batch-statistics BatchNorm)."""
    from gsn_amd import flags, layers
    m_rows, d_in, d_h, d_out = shape
    torch.manual_seed(5)
    m = layers.mlp(d_in, d_out, list(d_h), 0, act, bn).cuda().train()
    pbn = torch.nn.BatchNorm1d(d_out).cuda().train() if post else None
    pst = (pbn, "relu") if post else None
    for mod in list(m.bn) + ([pbn] if post else []):
        torch.nn.init.uniform_(mod.weight, 0.5, 1.5)
        torch.nn.init.normal_(mod.bias, 0.0, 0.2)
    x = torch.randn(m_rows, d_in, device="cuda", requires_grad=True)
    gy = torch.randn(m_rows, d_out, device="cuda")
    params = list(m.parameters()) + (list(pbn.parameters()) if post else [])
    state = [b.clone() for b in m.buffers()] + ([b.clone() for b in pbn.buffers()] if post else [])

    def run(native):
        flags.NATIVE_DENSE_BACKWARD = native
        for b, s0 in zip(list(m.buffers()) + (list(pbn.buffers()) if post else []), state):
            b.copy_(s0)
        for p in params + [x]:
            p.grad = None
        y = m(x, post=pst)
        (y * gy).sum().backward()
        return y.detach().clone(), [p.grad.clone() for p in params], x.grad.clone(), [b.clone() for b in m.buffers()]
    try:
        y1, g1, gx1, b1 = run(True)
        y0, g0, gx0, b0 = run(False)
    finally:
        flags.NATIVE_DENSE_BACKWARD = True
    scale = lambda t: max(float(t.abs().max()), 1e-6)
    assert float((y1 - y0).abs().max()) <= 2e-5 * scale(y0)
    gmax = max(scale(t) for t in g0)
    for a, b_ in zip(g1, g0):
        # (biases in front of a train-mode BatchNorm have zero gradient analytically: compare on the scale of the largest one)
        assert float((a - b_).abs().max()) <= 2e-4 * max(scale(b_), 1e-2 * gmax), (a.shape, float((a - b_).abs().max()), scale(b_))
    assert float((gx1 - gx0).abs().max()) <= 2e-4 * scale(gx0)
    for a, b_ in zip(b1, b0):
        assert torch.allclose(a.float(), b_.float(), rtol=1e-4, atol=1e-6)


def test_training_paths_use_native_adjoints():
    """In train mode no layer kind falls back to the PyTorch twin: general -> _general_train, gin / ogb -> propagate +
    native mlp; gradients reach every parameter and the inputs."""
    from gsn_amd import flags, layers, synth
    b = synth.zinc_shape_batch(6, seed=3)
    ei = torch.from_numpy(b.edge_index).cuda()
    n, E = b.num_nodes, b.num_edges
    base = dict(d_degree=1, degree_as_tag=False, retain_features=True, seed=0, activation_name="relu", bn=True, flow="source_to_target")
    cases = [layers.GSN_edge_sparse(d_in=8, d_ef=3, d_id=5, id_scope="local", d_msg=16, d_up=16, d_h=[16], msg_kind="general", **base),
             layers.GSN_sparse(d_in=8, d_id=5, id_scope="global", d_msg=None, d_up=16, d_h=[16], msg_kind="gin", train_eps=True,
                               id_embedding="one_hot_encoder", extend_dims=True, **base),
             layers.GSN_edge_sparse_ogb(d_in=8, d_ef=8, d_id=8, id_scope="local", d_msg=None, d_up=8, d_h=[16], msg_kind="ogb",
                                        train_eps=True, **base)]
    twin_calls = []
    orig = layers._HipWithTorchBackward.apply
    layers._HipWithTorchBackward.apply = staticmethod(lambda *a, **k: (twin_calls.append(1), orig(*a, **k))[1])
    try:
        for layer in cases:
            layer = layer.cuda().train()
            x = torch.randn(n, 8, device="cuda", requires_grad=True)
            kw = {"degrees": torch.zeros(n, device="cuda")}
            if isinstance(layer, layers.GSN_sparse):
                kw["identifiers"] = torch.randn(n, 5, device="cuda")
            elif isinstance(layer, layers.GSN_edge_sparse_ogb):
                kw["identifiers"] = torch.randn(E, 8, device="cuda"); kw["edge_features"] = torch.randn(E, 8, device="cuda")
            else:
                kw["identifiers"] = torch.randn(E, 5, device="cuda"); kw["edge_features"] = torch.randn(E, 3, device="cuda")
            y = layer(x, ei, **kw)
            y.square().sum().backward()
            assert x.grad is not None and torch.isfinite(x.grad).all()
            missing = [k for k, p in layer.named_parameters() if p.grad is None]
            assert not missing, missing
    finally:
        layers._HipWithTorchBackward.apply = orig
    assert not twin_calls, "a training-mode layer went through the PyTorch twin"


def test_full_size_properties_of_the_headline_layer():
    """BASELINE full size (65 536 ZINC-shaped graphs, E = 3.1 M): properties that need no reference.
    (i) a batch is a disjoint union: the forward of the first 1000 graphs alone equals those rows of the full forward;
    (ii) edge order is irrelevant: permuting the columns of edge_index (and the per-edge inputs with them) leaves the
    output unchanged up to fp32 summation order;  (iii) the integer-coded inputs give the same result."""
    import bench
    from gsn_amd import flags, layers
    b = bench.make_batch(65536, 3)
    dev = "cuda"
    N, E = b.num_nodes, b.num_edges
    ei = torch.from_numpy(b.edge_index).to(dev)
    xc = layers.Codes(torch.from_numpy(b.atom_type).to(dev), [28])
    ec = layers.Codes(torch.from_numpy(b.bond_type).to(dev), [4])
    ic = layers.Codes(torch.randint(0, 3, (E, 4), device=dev), [3, 3, 3, 3])
    x, idf, ef = xc.dense(), ic.dense(), ec.dense()
    deg = torch.zeros(N, device=dev)
    torch.manual_seed(0)
    layer = layers.GSN_edge_sparse(**bench.CTOR).to(dev).eval()
    with torch.no_grad():
        y = layer(x, ei, identifiers=idf, degrees=deg, edge_features=ef)
        n1, e1 = int(b.node_ptr[1000]), int(b.edge_ptr[1000])
        y1 = layer(x[:n1].contiguous(), ei[:, :e1].contiguous(), identifiers=idf[:e1].contiguous(), degrees=deg[:n1],
                   edge_features=ef[:e1].contiguous())
        perm = torch.randperm(E, device=dev)
        yp = layer(x, ei[:, perm].contiguous(), identifiers=idf[perm].contiguous(), degrees=deg, edge_features=ef[perm].contiguous())
        flags.CODE_STATUS_CHECK = True
        yc = layer(xc, ei, identifiers=ic, degrees=deg, edge_features=ec)
    scale = float(y.abs().max())
    assert torch.isfinite(y).all() and y.shape == (N, 128)
    assert float((y[:n1] - y1).abs().max()) <= 1e-5 * scale
    assert float((yp - y).abs().max()) <= 1e-5 * scale
    assert float((yc - y).abs().max()) <= 1e-5 * scale


def test_chain_shape_cases_reach_every_kernel_variant():
    """The shape cases above are only worth something if each of them runs on the kernel it names: run them in a child process
    with GSN_CHAIN_TRACE=1 and check that all six chain kernels (fp32 generic / role-split / stage-pipelined, their bf16x6
    twins) and the statistics pass were launched."""
    import os, subprocess, sys
    env = dict(os.environ, GSN_CHAIN_TRACE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-s", "-m", "gpu", "-k", "chain_kernel_shapes or fused_scatter"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    seen = {line.split()[1] for line in (r.stdout + r.stderr).splitlines() if line.startswith("gsn_chain_launch ")}
    for kernel in ("mlp_chain_kernel", "mlp_chain_kernel(stats)", "mlp_chain1_seg_kernel", "mlp_chain1_seg_bf16_kernel",
                   "mlp_chain2_pipe_kernel", "mlp_chain2_pipe_bf16_kernel"):
        assert kernel in seen, (kernel, sorted(seen))


@pytest.mark.parametrize("mix", ["exact", "mixed"])
def test_fused_scatter_add_with_bf16_exact_inputs(mix):
    """One-hot / small-integer inputs are exact in bf16: the bf16x6 edge stage then skips the plane products that read the
    (all-zero) middle and low planes of a tile.  'mixed': only part of the edge rows carry inexact values, so exact and
    inexact tiles alternate inside one launch."""
    from gsn_amd.layers import _Stage, run_stages, _csr_for
    torch.manual_seed(5)
    dev = "cuda"
    N, E = 900, 7000
    tgt = torch.randint(0, N, (E,), device=dev); src = torch.randint(0, N, (E,), device=dev)
    ei = torch.stack([src, tgt], 0)
    x = torch.nn.functional.one_hot(torch.randint(0, 28, (N,), device=dev), 28).float()
    ids = torch.randint(0, 5, (E, 12), device=dev).float()                 # small integers: exact as well
    ef = torch.nn.functional.one_hot(torch.randint(0, 4, (E,), device=dev), 4).float()
    if mix == "mixed":
        ef = ef.clone(); ef[tgt < N // 3] = torch.randn(int((tgt < N // 3).sum()), 4, device=dev)
    csr = _csr_for(ei, 1, N)
    K = 28 + 28 + 12 + 4
    W = torch.randn(128, K, device=dev) / K ** 0.5; b = torch.randn(128, device=dev)
    st = _Stage(W, b, None, "relu", [(x, csr.tgt), (x, csr.src), (ids, csr.perm), (ef, csr.perm)])
    out = run_stages([st], E, False, csr=csr)
    msg = torch.relu(torch.cat([x[tgt], x[src], ids, ef], 1).double() @ W.double().T + b.double())
    ref = torch.zeros(N, 128, dtype=torch.float64, device=dev).index_add_(0, tgt, msg)
    assert rel_err(out.double(), ref) < TOL


def test_csr_of_a_collated_batch_in_one_launch():
    """gsn_csr_build_graphs_hip (one wave per graph, sorted in LDS) against the generic gsn_csr_build_hip: identical seg_ptr,
    perm, sorted targets and sources for both rows of edge_index; duplicates, self loops, empty graphs, vertices without
    columns; pointers that do not describe the batch are reported; graphs beyond the LDS bound fall back."""
    import numpy as np
    from gsn_amd import flags, layers, synth
    rng = np.random.default_rng(3)
    graphs = [synth.zinc_shape_graph(rng) for _ in range(700)]
    graphs[5] = (4, np.zeros((2, 0), np.int64))                                    # no columns
    graphs[9] = (6, np.array([[0, 1, 1, 1, 2, 2, 5, 5], [1, 0, 0, 1, 2, 1, 0, 0]]))   # duplicates, self loops
    graphs[40] = synth.er_graph(300, 2500, 1)
    b = synth.collate(graphs)
    dev = torch.device("cuda")
    ei = torch.from_numpy(b.edge_index).to(dev)
    npt, ept = torch.from_numpy(b.node_ptr).to(dev), torch.from_numpy(b.edge_ptr).to(dev)
    mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
    N = b.num_nodes
    for row in (0, 1):
        want = layers.build_csr(ei[row], N, with_targets=True, other=ei[1 - row])
        got = layers.build_csr_graphs(ei[row], N, npt, ept, mn, me, other=ei[1 - row])
        for a, c in zip(want, got):
            assert torch.equal(a, c)
    # wrong pointers: a column leaves its graph -> reported
    bad = ept.clone()
    bad[3] += 2
    with pytest.raises(ValueError):
        layers.build_csr_graphs(ei[1], N, npt, bad, mn, me + 2, other=ei[0])
    with pytest.raises(ValueError):                                   # under-declared bound
        layers.build_csr_graphs(ei[1], N, npt, ept, mn, 100, other=ei[0])
    # pointers that leave the batch: reported, and nothing is written outside the arrays -- every seg_ptr entry stays in
    # [0, E] also when the status word is not read (check=False, what bench.py does)
    E = b.num_edges
    for which, delta in (("node_end", 7), ("edge_end", 5), ("edge_short", -3), ("edge_start", 2), ("node_back", None)):
        np2, ep2 = npt.clone(), ept.clone()
        if which == "node_end":
            np2[-1] += delta
        elif which == "edge_end":
            ep2[-1] += delta
        elif which == "edge_short":
            ep2[-1] += delta
        elif which == "edge_start":
            ep2[0] += delta
        else:
            np2[10] = np2[9] - 3                                       # not monotone
        with pytest.raises(ValueError):
            layers.build_csr_graphs(ei[1], N, np2, ep2, mn + 8, me + 8, other=ei[0])
        out = layers.build_csr_graphs(ei[1], N, np2, ep2, mn + 8, me + 8, other=ei[0], check=False)
        torch.cuda.synchronize()
        assert int(out[0].min()) >= 0 and int(out[0].max()) <= E, which
    assert layers.set_graph_partition(ei, npt, ept, 40000, 100000) is False      # beyond the LDS bound: generic build
    # a layer forward is the same with and without the partition
    torch.manual_seed(0)
    layer = layers.GSN_edge_sparse(d_in=8, d_ef=4, d_id=4, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=32,
                                   d_up=32, d_h=[32], seed=0, activation_name="relu", bn=False, msg_kind="general", flow="source_to_target").to(dev).eval()
    g = torch.Generator().manual_seed(1)
    x, ids, ef = torch.randn(N, 8, generator=g).to(dev), torch.randn(b.num_edges, 4, generator=g).to(dev), torch.randn(b.num_edges, 4, generator=g).to(dev)
    with torch.no_grad():
        y0 = layer(x, ei, identifiers=ids, degrees=torch.zeros(N, device=dev), edge_features=ef)
        layers._CSR_CACHE.clear()
        assert layers.set_graph_partition(ei, npt, ept, mn, me) is True
        y1 = layer(x, ei, identifiers=ids, degrees=torch.zeros(N, device=dev), edge_features=ef)
    assert torch.equal(y0, y1)
    layers._PARTITION.clear()


@pytest.mark.parametrize("cls,scope,bn,d", [("MPNN_edge_sparse", None, True, 128), ("GSN_edge_sparse", "local", True, 128),
                                             ("GSN_edge_sparse", "global", False, 96), ("GSN_sparse", "global", True, 128),
                                             ("MPNN_sparse", None, True, 200)])
def test_wide_edge_stage_split_into_node_product_and_gather_sum(cls, scope, bn, d, monkeypatch):
    """Layers 1.. of a d = 128 model (edge rows K = 260): cat(x_i, x_j, z) W^T = x_i W_i^T + x_j W_j^T + z W_z^T -- node
    product + gsn_edge_split_sum_hip -- against the fp32 oracle at 1e-5 element-wise, and against the E-row product path."""
    from gsn_amd import flags, layers, synth
    from oracle import oracle
    b = synth.zinc_shape_batch(300, seed=31)
    N, E = b.num_nodes, b.num_edges
    ctor = dict(d_in=d, d_degree=1, degree_as_tag=False, retain_features=True, d_msg=d, d_up=d, d_h=[d], seed=0,
                activation_name="relu", bn=bn, msg_kind="general", flow="source_to_target")
    if "edge" in cls:
        ctor["d_ef"] = 4
    if cls.startswith("GSN"):
        ctor.update(d_id=8, id_scope=scope)
    torch.manual_seed(3)
    layer = getattr(layers, cls)(**ctor)
    g = torch.Generator().manual_seed(4)
    for mod in layer.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.copy_(torch.rand(mod.running_mean.shape, generator=g) * 0.4 - 0.2)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
    layer.eval()
    x = torch.randn(N, d, generator=g)
    ids = torch.randn(N if scope == "global" else E, 8, generator=g) if cls.startswith("GSN") else None
    ef = torch.randn(E, 4, generator=g) if "edge" in cls else None
    ei = torch.from_numpy(b.edge_index)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    ref = oracle.layer_forward(cls, ctor, sd, x, ei, identifiers=ids, degrees=None, edge_features=ef, training=False)
    layer.cuda()
    kw = dict(identifiers=None if ids is None else ids.cuda(), degrees=torch.zeros(N, device="cuda"), edge_features=None if ef is None else ef.cuda())
    calls = []
    orig = layers._SparseLayer._split_edge_stage

    def spy(self, *a, **k):
        r = orig(self, *a, **k)
        calls.append(r is not None)
        return r
    monkeypatch.setattr(layers._SparseLayer, "_split_edge_stage", spy)
    monkeypatch.setattr(flags, "FUSED_LAYER", False)      # (d = 128 goes to the one-launch kernel of csrc/layer_w.hip otherwise: tests/test_fused_gpu.py)
    with torch.no_grad():
        y = layer(x.cuda(), ei.cuda(), **kw).cpu()
        assert calls == [True]
        monkeypatch.setattr(flags, "SPLIT_EDGE_STAGE", False)
        layers._CSR_CACHE.clear()
        y_old = layer(x.cuda(), ei.cuda(), **kw).cpu()
    assert elementwise_ok(y, ref), float((y - ref).abs().max() / ref.abs().max())
    assert elementwise_ok(y, y_old)


def test_edge_less_training_batch_through_batchnorm():
    """an edge-less batch in TRAIN mode through a `general` layer whose msg_fn has BatchNorm -- nn.BatchNorm1d takes a [0, C]
    input (empty output, running statistics untouched, the batch counted), so the layer must not raise: zero aggregates, zero gradients for
    msg_fn, update_fn trained on [x | 0]."""
    from gsn_amd import flags, layers
    from oracle import oracle
    ctor = dict(d_in=6, d_ef=3, d_id=4, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=16, d_up=16,
                d_h=[16], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target")
    torch.manual_seed(0)
    layer = layers.GSN_edge_sparse(**ctor).train()
    n = 7
    x = torch.randn(n, 6)
    ei = torch.zeros((2, 0), dtype=torch.int64)
    ids, ef = torch.zeros((0, 4)), torch.zeros((0, 3))
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    ref = oracle.layer_forward("GSN_edge_sparse", ctor, sd, x, ei, training=True, identifiers=ids, degrees=None, edge_features=ef)
    layer.cuda()
    rm0 = layer.msg_fn.bn[0].running_mean.clone()
    nbt0 = int(layer.msg_fn.bn[0].num_batches_tracked)
    y = layer(x.cuda(), ei.cuda(), identifiers=ids.cuda(), degrees=torch.zeros(n, device="cuda"), edge_features=ef.cuda())
    y.sum().backward()
    assert elementwise_ok(y.detach().cpu(), ref.detach())
    assert torch.equal(layer.msg_fn.bn[0].running_mean, rm0) and int(layer.msg_fn.bn[0].num_batches_tracked) == nbt0 + 1
    for name, p in layer.msg_fn.named_parameters():
        assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
    assert any(p.grad is not None and float(p.grad.abs().max()) > 0 for p in layer.update_fn.parameters())


def test_eval_after_training_uses_the_updated_running_statistics():
    """eval forward (fills the eval-mode BatchNorm vector cache) -> train forward (the finalize kernel writes running_mean / running_var through
    raw pointers) -> eval forward: must see the NEW running statistics (their version counters are moved with the kernel's write)."""
    from gsn_amd import flags, layers
    from oracle import oracle
    ctor = dict(d_in=6, d_ef=3, d_id=4, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=16, d_up=16,
                d_h=[16], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target")
    torch.manual_seed(1)
    g = torch.Generator().manual_seed(2)
    n, E = 40, 160
    x, ei = torch.randn(n, 6, generator=g), torch.randint(0, n, (2, E), generator=g)
    ids, ef = torch.randn(E, 4, generator=g), torch.randn(E, 3, generator=g)
    layer = layers.GSN_edge_sparse(**ctor).cuda()
    kw = dict(identifiers=ids.cuda(), degrees=torch.zeros(n, device="cuda"), edge_features=ef.cuda())
    with torch.no_grad():
        layer.eval()
        layer(x.cuda(), ei.cuda(), **kw)
        layer.train()
        v0 = layer.msg_fn.bn[0].running_mean._version
        for _ in range(3):
            layer(x.cuda() * 3.0 + 1.0, ei.cuda(), **kw)
        assert layer.msg_fn.bn[0].running_mean._version > v0
        layer.eval()
        y = layer(x.cuda(), ei.cuda(), **kw)
    sd = {k: v.detach().cpu().clone() for k, v in layer.state_dict().items()}
    ref = oracle.layer_forward("GSN_edge_sparse", ctor, sd, x, ei, training=False, identifiers=ids, degrees=None, edge_features=ef)
    assert elementwise_ok(y.cpu(), ref), rel_err(y.cpu(), ref)


def test_segment_sums_of_a_column_slice():
    """gsn_segment_sum_rows_hip (layers._segment_sum_cols): per-vertex sums of rows[:, col0 : col0 + width] through edge_index[mode], read in
    place (row stride = the full row), against index_add_ in fp64; short segments, a hub, isolated vertices, both modes, several slices."""
    from gsn_amd.layers import _segment_sum_cols, _CSR_CACHE
    g = torch.Generator().manual_seed(11)
    for n, E, K in ((1, 0, 8), (50, 170, 272), (3000, 9000, 272), (40, 4000, 36), (20000, 50000, 140)):
        ei = torch.randint(0, n, (2, E), generator=g)
        if E:
            ei[1, : E // 3] = 0                                 # a hub
        rows = torch.randn(E, K, generator=g)
        eig, rg = ei.cuda(), rows.cuda()
        _CSR_CACHE.clear()
        for mode in (0, 1):
            for col0, width in {(0, K), (4, min(12, K - 4)), (K - 8, 8), (max(K - 132, 0), min(128, K))}:
                got = _segment_sum_cols(eig, mode, n, rg, col0, width).cpu()
                ref = torch.zeros(n, width, dtype=torch.float64).index_add_(0, ei[mode], rows[:, col0:col0 + width].double())
                assert got.shape == (n, width)
                assert float((got.double() - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1.0), (n, E, K, mode, col0, width)

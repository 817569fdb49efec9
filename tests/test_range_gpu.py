"""Dynamic range and non-finite values through the split-operand matrix kernels on a real MI355X.

The dense kernels do not multiply fp32 numbers on fp32 hardware: the chain / linear kernels split each operand into three bf16
planes (six plane products), the one-launch layer kernel into two fp16 planes after a power-of-two row / matrix scaling (three
plane products).  These tests pin what that means outside the comfortable range of `randn` data:
  * rows mixing magnitudes from 1e-30 to 1e30, fp32 subnormals, signed zeros  -> element-wise
        |got - ref| <= 1e-5 |ref| + 1e-5 * sum_k |x_k| |w_k|      (ref: an fp32 torch matmul; the floor is the product's own
    condition scale, computed in fp64 -- an element that is small only through cancellation cannot be asked for more)
  * Inf / NaN inputs: the row that holds one comes out NaN in all its columns (documented deviation: fp32 arithmetic keeps a
    signed Inf where no Inf - Inf arises); every other row is bit-identical to the run without it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _wide_rows(m, k, seed, lo=-30, hi=30, sparse=0.0):
    g = torch.Generator().manual_seed(seed)
    mant = torch.rand(m, k, generator=g) + 1.0
    expo = torch.randint(lo, hi + 1, (m, k), generator=g).float()
    sign = torch.where(torch.rand(m, k, generator=g) < 0.5, -1.0, 1.0)
    x = sign * mant * torch.pow(torch.tensor(10.0), expo)
    if sparse:
        x = torch.where(torch.rand(m, k, generator=g) < sparse, torch.zeros(()), x)
    return x.float()


def _check(got, x, w, b, act=None, rtol=1e-5):
    ref = torch.nn.functional.linear(x, w, b)                                     # fp32 torch matmul
    scale = (x.double().abs() @ w.double().abs().t()) + (b.double().abs() if b is not None else 0.0)
    if act == "relu":
        ref = torch.relu(ref)
    finite = torch.isfinite(ref)
    err = (got.double() - ref.double()).abs()
    bound = rtol * ref.double().abs() + rtol * scale
    bad = (err > bound) & finite
    assert not bool(bad.any()), "worst excess %.3g at %s" % (float((err / bound)[finite].max()), str(torch.nonzero(bad)[:3].tolist()))
    assert bool(torch.isfinite(got)[finite].all())


@pytest.mark.parametrize("k,n", [(72, 128), (160, 128), (260, 128), (300, 600)])
def test_bf16x6_kernels_wide_dynamic_range(k, n):
    """gsn_mlp_chain_fwd_hip (K <= 160) and gsn_linear_fwd_hip (any K): rows mixing 1e-30 .. 1e30 with well-scaled weights."""
    from gsn_amd import layers
    m = 777
    x = _wide_rows(m, k, seed=k, lo=-30, hi=30, sparse=0.1)
    g = torch.Generator().manual_seed(1)
    w = torch.randn(n, k, generator=g) * 0.1
    b = torch.randn(n, generator=g)
    xs, ws, bs = x.cuda(), w.cuda(), b.cuda()
    st = layers._Stage(ws, bs, None, "identity", [(xs, None)])
    y = layers.run_stages([st], m, False)
    _check(y.cpu(), x, w, b)
    # columns given as two blocks, the second gathered through an index (the gather / concatenation path)
    if k % 8 == 0:
        idx = torch.randperm(m, generator=g)
        second = torch.empty(m, k - k // 2)
        second[idx] = x[:, k // 2:]
        st = layers._Stage(ws, bs, None, "relu", [(xs[:, :k // 2].contiguous(), None), (second.cuda(), idx.cuda())])
        y = layers.run_stages([st], m, False)
        _check(y.cpu(), x, w, b, act="relu")


def test_bf16x6_kernels_subnormals_and_signed_zero():
    from gsn_amd import layers
    m, k, n = 256, 96, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(m, k, generator=g)
    x[::3] *= 1e-41                                         # fp32 subnormals
    x[1::7, ::2] = -0.0
    w = torch.randn(n, k, generator=g)
    w[::5] *= 1e-39
    y = layers.run_stages([layers._Stage(w.cuda(), None, None, "identity", [(x.cuda(), None)])], m, False)
    _check(y.cpu(), x, w, None)


@pytest.mark.parametrize("k,n", [(160, 128), (260, 128)])
def test_dense_kernels_non_finite_rows(k, n):
    """A row with an Inf or a NaN input never produces a finite number: every element is the signed Inf fp32 arithmetic gives
    (the fp32-MFMA chain kernel) or NaN (the split-operand kernels: Inf - Inf inside the split); all other rows are untouched."""
    from gsn_amd import layers
    m = 300
    g = torch.Generator().manual_seed(4)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) * 0.1
    clean = layers.run_stages([layers._Stage(w.cuda(), None, None, "identity", [(x.cuda(), None)])], m, False).cpu()
    x2 = x.clone()
    x2[5, 7] = float("inf"); x2[40, 0] = float("-inf"); x2[41, k - 1] = float("nan")
    y = layers.run_stages([layers._Stage(w.cuda(), None, None, "identity", [(x2.cuda(), None)])], m, False).cpu()
    ref = torch.nn.functional.linear(x2, w)
    bad = torch.zeros(m, dtype=torch.bool); bad[[5, 40, 41]] = True
    assert not bool(torch.isfinite(y[bad]).any())
    same_inf = torch.isinf(y[bad]) & (y[bad] == ref[bad])
    assert bool((same_inf | torch.isnan(y[bad])).all())
    assert torch.equal(y[~bad], clean[~bad])


CTOR = dict(d_in=28, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=128,
            d_up=128, d_h=[128], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target")


def _layer_and_batch(n_graphs, seed):
    from gsn_amd import layers, synth
    b = synth.zinc_shape_batch(n_graphs, seed=seed)
    torch.manual_seed(seed)
    layer = layers.GSN_edge_sparse(**CTOR)
    g = torch.Generator().manual_seed(seed + 1)
    for mod in layer.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.copy_(torch.rand(mod.running_mean.shape, generator=g) * 0.4 - 0.2)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
    layer.eval()
    return layer, b


def _fused(layer, x, ei, ids, ef, capfd=None):
    import os
    from gsn_amd import layers
    layers._CSR_CACHE.clear()
    os.environ["GSN_CHAIN_TRACE"] = "1"
    try:
        with torch.no_grad():
            y = layer(x.cuda(), ei.cuda(), identifiers=ids.cuda(), degrees=torch.zeros(x.shape[0], device="cuda"), edge_features=ef.cuda())
            torch.cuda.synchronize()
    finally:
        os.environ.pop("GSN_CHAIN_TRACE", None)
    if capfd is not None:
        assert "layer_fused_kernel" in capfd.readouterr().err
    return y.cpu()


@pytest.mark.parametrize("scale_x,scale_e", [(1e-20, 1.0), (1e15, 1e-10), (1.0, 1e18), (3e-38, 3e-38)])
def test_fused_layer_row_scaling_over_the_exponent_range(scale_x, scale_e, capfd):
    """The fp16x3 layer kernel: inputs whose magnitudes sit far outside fp16's range (and differ by orders of magnitude between
    the blocks of one edge row) against the fp32 oracle, element-wise with a floor of 1e-5 of the row's largest output."""
    from oracle import oracle
    layer, b = _layer_and_batch(200, seed=9)
    g = torch.Generator().manual_seed(10)
    x = torch.randn(b.num_nodes, 28, generator=g) * scale_x
    ids = torch.randn(b.num_edges, 12, generator=g) * scale_e
    ef = torch.randn(b.num_edges, 4, generator=g) * scale_e * 1e3
    ei = torch.from_numpy(b.edge_index)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    ref = oracle.layer_forward("GSN_edge_sparse", CTOR, sd, x, ei, identifiers=ids, degrees=None, edge_features=ef, training=False)
    y = _fused(layer.cuda(), x, ei, ids, ef, capfd)
    floor = 1e-5 * ref.abs().amax(dim=1, keepdim=True)
    assert bool(((y - ref).abs() <= 1e-5 * ref.abs() + floor).all()), float(((y - ref).abs() / (ref.abs().amax(dim=1, keepdim=True) + 1e-45)).max())


def test_fused_layer_non_finite_rows(capfd):
    """An Inf / NaN in a node row or an edge row: exactly the nodes that see it (the node itself; the targets of the edge row, and
    of every edge that gathers the node) come out NaN in all columns; every other output row equals the clean run."""
    layer, b = _layer_and_batch(64, seed=12)
    N, E = b.num_nodes, b.num_edges
    x = torch.nn.functional.one_hot(torch.from_numpy(b.atom_type), 28).float()
    ef = torch.nn.functional.one_hot(torch.from_numpy(b.bond_type), 4).float()
    ids = (torch.rand(E, 12, generator=torch.Generator().manual_seed(2)) < 0.3).float()
    ei = torch.from_numpy(b.edge_index)
    layer.cuda()
    clean = _fused(layer, x, ei, ids, ef)
    assert bool(torch.isfinite(clean).all())
    src, tgt = ei[0], ei[1]
    for kind in ("x_inf", "x_nan", "edge_inf", "id_nan"):
        x2, ef2, ids2 = x.clone(), ef.clone(), ids.clone()
        expect = torch.zeros(N, dtype=torch.bool)
        if kind.startswith("x"):
            v = 17
            x2[v, 3] = float("inf") if kind == "x_inf" else float("nan")
            expect[v] = True                                  # its own row of the node stage
            expect[tgt[(src == v) | (tgt == v)]] = True        # every edge row that gathers x_v (as x_j or as x_i) poisons its target
        else:
            e = 33
            if kind == "edge_inf":
                ef2[e, 1] = float("-inf")
            else:
                ids2[e, 5] = float("nan")
            expect[tgt[e]] = True
        y = _fused(layer, x2, ei, ids2, ef2, capfd)
        assert bool(torch.isnan(y[expect]).all()), kind
        # (rows that share a 64-row chunk with the poisoned row take the three-product path instead of the two-product one: same
        # terms, another summation order)
        ok = y[~expect]
        assert bool(torch.isfinite(ok).all()), kind
        assert bool(((ok - clean[~expect]).abs() <= 2e-6 * clean[~expect].abs().amax(dim=1, keepdim=True)).all()), kind


def test_fused_layer_non_finite_weight():
    layer, b = _layer_and_batch(8, seed=13)
    x = torch.nn.functional.one_hot(torch.from_numpy(b.atom_type), 28).float()
    ef = torch.nn.functional.one_hot(torch.from_numpy(b.bond_type), 4).float()
    ids = torch.zeros(b.num_edges, 12)
    ei = torch.from_numpy(b.edge_index)
    layer.cuda()
    layer.update_fn.fc[1].weight.data[3, 4] = float("inf")
    from gsn_amd import layers
    layers.invalidate_caches(layer)
    y = _fused(layer, x, ei, ids, ef)
    assert bool(torch.isnan(y).all())

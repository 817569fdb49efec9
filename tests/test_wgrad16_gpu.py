"""gsn_wgrad_f16x3_hip (csrc/wgrad_f16.hip): the weight gradient of a dense stage, grad_w += gH^T X (torch.nn.Linear's weight.grad under
models_misc.py:52-58), from the fp16 planes the f16x3 products of a training step leave behind -- against float64, beside the bf16x6 kernel
(gsn_wgrad_hip) and an fp32 product; ragged row counts, widths that are not whole K slices, rows of very different magnitude inside a slab,
zero rows, a non-finite row, both operands' pre-pass by itself (gsn_linear_f16x3_split_rows_hip)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _split(t):
    from gsn_amd import _abi
    m, k = t.shape
    L = _abi.lib()
    scratch = torch.empty(int(L.gsn_linear_f16x3_scratch_bytes(m, k)), dtype=torch.uint8, device=t.device)
    arr = (_abi.gsn_block * 1)()
    arr[0].data = t.data_ptr(); arr[0].idx = None; arr[0].idx32 = None; arr[0].width = k
    _abi.check(L.gsn_linear_f16x3_split_rows_hip(m, 1, arr, scratch.data_ptr(), _abi.current_stream()), "gsn_linear_f16x3_split_rows_hip")
    return scratch


def _wgrad16(gh, x):
    from gsn_amd import _abi
    m, n_out = gh.shape
    k = x.shape[1]
    sg, sx = _split(gh), _split(x)
    gw = torch.zeros(n_out, k, device=gh.device)
    _abi.check(_abi.lib().gsn_wgrad_f16x3_hip(m, n_out, k, sg.data_ptr(), sx.data_ptr(), gw.data_ptr(), _abi.current_stream()), "gsn_wgrad_f16x3_hip")
    return gw


def _wgrad_bf16(gh, x):
    from gsn_amd import _abi
    m, n_out = gh.shape
    gw = torch.zeros(n_out, x.shape[1], device=gh.device)
    arr = (_abi.gsn_block * 1)()
    arr[0].data = x.data_ptr(); arr[0].idx = None; arr[0].idx32 = None; arr[0].width = x.shape[1]
    _abi.check(_abi.lib().gsn_wgrad_hip(m, n_out, gh.data_ptr(), 1, arr, gw.data_ptr(), _abi.current_stream()), "gsn_wgrad_hip")
    return gw


def _errs(gw, gh, x):
    ref = gh.double().t() @ x.double()
    bound = gh.double().abs().t() @ x.double().abs()
    return ((gw.double() - ref).abs() / bound.clamp_min(1e-300)).max().item(), ref, bound


@pytest.mark.parametrize("m_rows,n_out,k", [(5000, 600, 300), (105083, 300, 600), (777, 72, 164), (33, 300, 600), (100000, 128, 260), (17, 12, 8), (1, 4, 4),
                                            (4096, 600, 300), (257, 132, 36)])
def test_weight_gradient_f16x3_vs_fp64(m_rows, n_out, k):
    """Element-wise against sum_r |g||x| (what an fp32 accumulation of the same terms can promise), with column magnitudes over two decades and
    row magnitudes over six (gH) / four (X): not worse than twice an fp32 product's error (1e-6 where few rows leave nothing to average), and the
    bf16x6 kernel beside it."""
    torch.manual_seed(m_rows + n_out)
    rows_g = torch.logspace(-3, 3, m_rows, device="cuda")[torch.randperm(m_rows, device="cuda")].unsqueeze(1)
    rows_x = torch.logspace(-2, 2, m_rows, device="cuda")[torch.randperm(m_rows, device="cuda")].unsqueeze(1)
    gh = torch.randn(m_rows, n_out, device="cuda") * torch.logspace(-1, 1, n_out, device="cuda") * rows_g * 1e-6
    x = torch.randn(m_rows, k, device="cuda") * torch.logspace(-1, 1, k, device="cuda") * rows_x
    gw = _wgrad16(gh, x)
    err, ref, bound = _errs(gw, gh, x)
    err32 = (((gh.t() @ x).double() - ref).abs() / bound.clamp_min(1e-300)).max().item()
    err_bf = _errs(_wgrad_bf16(gh, x), gh, x)[0]
    assert err <= max(2.0 * err32, 1e-6), (err, err32, err_bf)


def test_values_far_below_their_rows_largest_keep_an_absolute_precision():
    """The documented limit of fp16 planes under ONE scale per row: a value v of a row whose largest magnitude is R is held to |v| 2^-22 or R 2^-39,
    whichever is larger (the low plane ends at fp16's smallest subnormal) -- column magnitudes over eight decades: every element within
    2^-21 sum|g||x| + 2^-38 sum_r max|g_r| max|x_r|."""
    torch.manual_seed(11)
    m, n_out, k = 2000, 128, 64
    gh = torch.randn(m, n_out, device="cuda") * torch.logspace(-4, 4, n_out, device="cuda")
    x = torch.randn(m, k, device="cuda") * torch.logspace(-4, 4, k, device="cuda")
    gw = _wgrad16(gh, x)
    ref = gh.double().t() @ x.double()
    bound = 2.0 ** -21 * (gh.double().abs().t() @ x.double().abs()) + 2.0 ** -38 * (gh.double().abs().amax(1) * x.double().abs().amax(1)).sum()
    assert ((gw.double() - ref).abs() <= bound).all(), ((gw.double() - ref).abs() / bound).max().item()


@pytest.mark.parametrize("m_rows,n_out,k", [(5000, 600, 300), (777, 72, 164), (105083, 300, 600), (17, 12, 8)])
def test_the_lds_dma_kernel_gives_the_same_gradient(m_rows, n_out, k, monkeypatch):
    """GSN_WGRAD16_DMA=1: whole lines straight into LDS (global_load_lds_dwordx4), operand fragments by transposing reads (ds_read_b64_tr_b16) --
    measured slower than the register-staged kernel and not the default; same planes, same products: the same tile up to the order of the atomic adds."""
    torch.manual_seed(m_rows)
    gh = torch.randn(m_rows, n_out, device="cuda") * torch.logspace(-3, 3, m_rows, device="cuda")[:, None] * 1e-5
    x = torch.randn(m_rows, k, device="cuda") * torch.logspace(-2, 2, m_rows, device="cuda").flip(0)[:, None]
    staged = _wgrad16(gh, x)
    monkeypatch.setenv("GSN_WGRAD16_DMA", "1")
    dma = _wgrad16(gh, x)
    err, ref, bound = _errs(dma, gh, x)
    assert err <= 1e-6, err
    assert torch.allclose(dma, staged, rtol=1e-5, atol=1e-6 * float(ref.abs().max()))


def test_rows_far_below_the_slab_maximum_and_zero_rows():
    """The reconciliation of the row scales: rows 2^-30 below the largest row of their slab vanish (their terms are below 2^-30 of the sum), zero
    rows of either operand add nothing, a slab of zero rows adds nothing; error against the sum of magnitudes as above."""
    torch.manual_seed(5)
    m, n_out, k = 3000, 256, 128
    gh = torch.randn(m, n_out, device="cuda")
    x = torch.randn(m, k, device="cuda")
    gh[::7] *= 2.0 ** -30
    x[3::11] *= 2.0 ** -12
    gh[100:140] = 0
    x[500:530] = 0
    gh[1024:2048] = 0                       # (whole slabs at 64 rows per slab)
    gw = _wgrad16(gh, x)
    err, ref, bound = _errs(gw, gh, x)
    assert err <= 6e-7, err
    z = _wgrad16(torch.zeros_like(gh), x)
    assert torch.count_nonzero(z) == 0
    z = _wgrad16(gh, torch.zeros_like(x))
    assert torch.count_nonzero(z) == 0


def test_a_non_finite_row_makes_its_tile_columns_nan_and_nothing_is_read_past_the_rows():
    torch.manual_seed(6)
    m, n_out, k = 1000, 132, 72
    gh = torch.randn(m, n_out, device="cuda")
    x = torch.randn(m, k, device="cuda")
    gh[999, 5] = float("inf")
    gw = _wgrad16(gh, x)
    assert torch.isnan(gw).any()
    gh[999, 5] = 1.0
    gw = _wgrad16(gh, x)
    assert torch.isfinite(gw).all()
    err = _errs(gw, gh, x)[0]
    assert err <= 6e-7, err


def test_added_to_not_overwritten():
    torch.manual_seed(7)
    gh, x = torch.randn(500, 64, device="cuda"), torch.randn(500, 96, device="cuda")
    once = _wgrad16(gh, x)
    from gsn_amd import _abi
    sg, sx = _split(gh), _split(x)
    gw = once.clone()
    _abi.check(_abi.lib().gsn_wgrad_f16x3_hip(500, 64, 96, sg.data_ptr(), sx.data_ptr(), gw.data_ptr(), _abi.current_stream()), "gsn_wgrad_f16x3_hip")
    assert torch.allclose(gw, 2 * once, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("d_out", [300, 302])
@pytest.mark.parametrize("train", [True, False])
def test_dense_stages_take_the_plane_kernel_and_match_float64_autograd(train, d_out, monkeypatch):
    """Linear -> BatchNorm -> ReLU -> Linear over 20 000 rows under autograd (models_misc.py:41-59; the d = 300 node stages of
    GSN_edge_sparse_ogb.py:63-129): both weight gradients come from gsn_wgrad_f16x3_hip -- the forward products' row scratches are kept, the
    input-gradient products leave gH's -- and every gradient agrees with a float64 PyTorch evaluation; with the switch off the same stages run
    on gsn_wgrad_hip.  The BatchNorm stage writes its output as planes only (gsn_bn_act_planes_hip) and its adjoint writes gH as planes only
    (gsn_bn_act_bwd_planes_hip).  d_out = 302: the second stage's gH cannot be split (width not a multiple of 4) -- its weight gradient falls back to
    gsn_wgrad_hip on fp32 rows that the forward pass did not keep: they are made again from the pre-BN rows."""
    from gsn_amd import _abi, _autograd, flags
    from gsn_amd._dense import _Stage
    torch.manual_seed(3)
    m, d, h = 20000, 300, 600
    x = (torch.randn(m, d, device="cuda") * torch.logspace(-1, 1, m, device="cuda")[:, None]).requires_grad_(True)
    lin1, bn, lin2 = torch.nn.Linear(d, h).cuda(), torch.nn.BatchNorm1d(h).cuda(), torch.nn.Linear(h, d_out).cuda()
    bn.train(train)
    if not train:
        with torch.no_grad():
            bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    wout = torch.randn(m, d_out, device="cuda")

    def run():
        for p in (*lin1.parameters(), *bn.parameters(), *lin2.parameters()):
            p.grad = None
        x.grad = None
        stages = [_Stage(lin1.weight, lin1.bias, bn, "relu", blocks=[(x, None)]), _Stage(lin2.weight, lin2.bias, None, "identity")]
        y = _autograd.run_stages_autograd(stages, m, train)
        (y * wout).sum().backward()
        return [t.grad.clone() for t in (x, lin1.weight, lin1.bias, bn.weight, bn.bias, lin2.weight, lin2.bias)]

    called = []
    real = _abi.check
    monkeypatch.setattr(_abi, "check", lambda rc, what="": (called.append(what), real(rc, what))[1])
    got = run()
    if d_out == 300:
        assert called.count("gsn_wgrad_f16x3_hip") == 2 and "gsn_wgrad_hip" not in called
    else:
        assert called.count("gsn_wgrad_f16x3_hip") == 1 and called.count("gsn_wgrad_hip") == 1 and called.count("gsn_bn_act_hip") == 1
    assert called.count("gsn_bn_act_planes_hip") == 1 and called.count("gsn_bn_act_bwd_planes_hip") == 1
    monkeypatch.setattr(flags, "WGRAD_F16X3", False)
    called.clear()
    old = run()
    assert called.count("gsn_wgrad_hip") == 2 and "gsn_wgrad_f16x3_hip" not in called
    # float64 reference
    x64 = x.detach().double().requires_grad_(True)
    l1, b64, l2 = torch.nn.Linear(d, h).cuda().double(), torch.nn.BatchNorm1d(h).cuda().double(), torch.nn.Linear(h, d_out).cuda().double()
    l1.load_state_dict(lin1.state_dict()); b64.load_state_dict(bn.state_dict()); l2.load_state_dict(lin2.state_dict())
    b64.train(train)
    (l2(torch.relu(b64(l1(x64)))) * wout.double()).sum().backward()
    ref = [x64.grad, l1.weight.grad, l1.bias.grad, b64.weight.grad, b64.bias.grad, l2.weight.grad, l2.bias.grad]
    for name, g, o, r in zip(("x", "W1", "b1", "gamma", "beta", "W2", "b2"), got, old, ref):
        scale = r.abs().max().item()
        e_new, e_old = (g.double() - r).abs().max().item() / scale, (o.double() - r).abs().max().item() / scale
        assert e_new <= max(2.0 * e_old, 2e-6), (name, e_new, e_old)


@pytest.mark.parametrize("m,k,n", [(70000, 300, 600), (66000, 600, 300), (131072, 64, 640)])
def test_product_on_128x320_tiles_matches_the_128x128_kernel(m, k, n, monkeypatch):
    """GSN_L16_WIDE=1: linear_f16x3_dma_kernel<5> (one workgroup per CU, 160 accumulator registers per wave; measured equal, not the default) --
    rows, column statistics and the relu epilogue against the default instantiation and float64 (models_misc.py:52-58)."""
    from gsn_amd import _abi
    from gsn_amd._dense import _f16x3_weights
    torch.manual_seed(m)
    L = _abi.lib()
    x = torch.randn(m, k, device="cuda") * torch.logspace(-2, 2, m, device="cuda")[:, None]
    W = torch.randn(n, k, device="cuda") / k ** 0.5
    b = torch.randn(n, device="cuda")
    planes, col_inv = _f16x3_weights(W, W)
    sc = _split(x)
    one = (_abi.gsn_block * 1)()
    one[0].data = x.data_ptr(); one[0].idx = None; one[0].idx32 = None; one[0].width = k

    def run():
        y, h = torch.empty(m, n, device="cuda"), torch.empty(m, n, device="cuda")
        st = torch.zeros(2, n, dtype=torch.float64, device="cuda")
        _abi.check(L.gsn_linear_f16x3_fwd_presplit_hip(m, 1, one, planes.data_ptr(), col_inv.data_ptr(), b.data_ptr(), n, None, None, None, 1, sc.data_ptr(),
                                                       y.data_ptr(), _abi.current_stream()), "product")
        _abi.check(L.gsn_linear_f16x3_fwd_stats_presplit_hip(m, 1, one, planes.data_ptr(), col_inv.data_ptr(), b.data_ptr(), n, sc.data_ptr(), h.data_ptr(),
                                                             st.data_ptr(), _abi.current_stream()), "product + statistics")
        return y, h, st
    y0, h0, st0 = run()
    monkeypatch.setenv("GSN_L16_WIDE", "1")
    y1, h1, st1 = run()
    assert torch.equal(y0, y1) and torch.equal(h0, h1)              # (same planes, same products in the same order per output element)
    assert torch.allclose(st0, st1, rtol=1e-12, atol=0)
    ref = x[:8192].double() @ W.double().t() + b.double()
    err = ((h1[:8192].double() - ref).abs() / ref.abs().amax(1, keepdim=True)).max().item()
    assert err <= 2e-6, err


def test_new_entries_refuse_what_they_do_not_take_and_accept_empty_calls():
    """Error behaviour of the r06 entries: unsupported shapes come back as GSN_E_UNSUPPORTED / GSN_E_INVALID with a message (never a wrong result), zero
    rows are a no-op (the reference's layers see edge-less and node-less batches: utils_data_gen.py:86-108)."""
    from gsn_amd import _abi
    L = _abi.lib()
    st = _abi.current_stream()
    h = torch.randn(64, 644, device="cuda"); g = torch.randn(64, 644, device="cuda")
    v = torch.ones(644, device="cuda")
    scr = torch.empty(int(L.gsn_linear_f16x3_scratch_bytes(64, 644)), dtype=torch.uint8, device="cuda")
    sums = torch.zeros(2, 644, dtype=torch.float64, device="cuda")
    # more than 640 columns / a width that is not a multiple of 4
    assert L.gsn_bn_act_planes_hip(64, 644, h.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr(), 1, None, scr.data_ptr(), st) == -2
    assert L.gsn_bn_act_planes_hip(64, 30, h.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr(), 1, None, scr.data_ptr(), st) == -2
    assert L.gsn_bn_act_bwd_planes_hip(64, 644, g.data_ptr(), h.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr(), 1, 1, sums.data_ptr(), scr.data_ptr(),
                                       None, st) == -2
    assert L.gsn_bn_act_bwd_planes_hip(64, 128, g.data_ptr(), h.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr(), 0, 1, sums.data_ptr(), scr.data_ptr(),
                                       None, st) == -2          # (train_bn 0: not a BatchNorm stage)
    # misaligned rows / scratch
    assert L.gsn_bn_act_planes_hip(64, 128, h.data_ptr() + 4, v.data_ptr(), v.data_ptr(), v.data_ptr(), 1, None, scr.data_ptr(), st) == -1
    assert L.gsn_wgrad_f16x3_hip(64, 128, 128, scr.data_ptr() + 4, scr.data_ptr(), g.data_ptr(), st) == -1
    assert b"aligned" in L.gsn_last_error()
    one = (_abi.gsn_block * 1)()
    one[0].data = h.data_ptr(); one[0].idx = None; one[0].idx32 = None; one[0].width = 30
    assert L.gsn_linear_f16x3_split_rows_hip(64, 1, one, scr.data_ptr(), st) == -2
    # zero rows: nothing launched, nothing touched
    gw = torch.full((8, 8), 3.0, device="cuda")
    assert L.gsn_wgrad_f16x3_hip(0, 8, 8, None, None, gw.data_ptr(), st) == 0
    assert L.gsn_bn_act_planes_hip(0, 128, None, None, None, None, 1, None, None, st) == 0
    assert L.gsn_bn_act_bwd_planes_hip(0, 128, None, None, None, None, None, None, 1, 1, None, None, None, st) == 0
    assert L.gsn_linear_f16x3_split_rows_hip(0, 1, one, None, st) == 0
    torch.cuda.synchronize()
    assert bool((gw == 3.0).all())
    assert L.gsn_linear_f16x3_mpad(0) == 0 and L.gsn_linear_f16x3_mpad(1) == 256 and L.gsn_linear_f16x3_mpad(128) == 256 and L.gsn_linear_f16x3_mpad(129) == 512


def test_mlp_on_an_empty_batch_and_on_one_row_under_the_plane_switches():
    """Zero rows and one row through run_stages_autograd with every r06 switch on: the plane paths do not apply (too few tiles), nothing breaks."""
    from gsn_amd import _autograd
    from gsn_amd._dense import _Stage
    lin1, bn, lin2 = torch.nn.Linear(300, 600).cuda(), torch.nn.BatchNorm1d(600).cuda(), torch.nn.Linear(600, 300).cuda()
    bn.eval()
    for m in (0, 1):
        x = torch.randn(m, 300, device="cuda", requires_grad=True)
        stages = [_Stage(lin1.weight, lin1.bias, bn, "relu", blocks=[(x, None)]), _Stage(lin2.weight, lin2.bias, None, "identity")]
        y = _autograd.run_stages_autograd(stages, m, False)
        assert y.shape == (m, 300)
        y.sum().backward()
        assert x.grad.shape == (m, 300) and torch.isfinite(lin1.weight.grad).all()
        if m == 1:
            ref = lin2(torch.relu(bn(lin1(x.detach()))))
            assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dma", ["0", "1"])
def test_two_million_rows_keep_the_slab_tables_inside_the_lds_budget(dma, monkeypatch):
    """M = 2 000 000 rows x 64 x 64 (one tile): the slab-length rule asks for more rows per slab than the scale tables may hold -- capped, more slabs
    instead; both kernels."""
    monkeypatch.setenv("GSN_WGRAD16_DMA", dma)
    torch.manual_seed(0)
    m = 2000000
    gh = torch.randn(m, 64, device="cuda") * 1e-3
    x = torch.randn(m, 64, device="cuda")
    gw = _wgrad16(gh, x)
    ref = (gh.t().double() @ x.double())
    assert float((gw.double() - ref).abs().max() / ref.abs().max()) <= 2e-5


def test_plane_paths_on_one_and_a_half_million_rows():
    """1 500 000 rows through Linear -> BatchNorm (train) -> elu -> Linear with the plane switches on and off: 64-bit row offsets in the plane kernels,
    the capped slab tables, the 2 048-workgroup regime of the slab rule -- same gradients as the fp32-row passes."""
    from gsn_amd import _autograd, flags
    from gsn_amd._dense import _Stage
    torch.manual_seed(1)
    m = 1500000
    x = torch.randn(m, 128, device="cuda")
    lin1, bn, lin2 = torch.nn.Linear(128, 256).cuda(), torch.nn.BatchNorm1d(256).cuda(), torch.nn.Linear(256, 132).cuda()
    wout = torch.randn(m, 132, device="cuda")
    params = [*lin1.parameters(), *bn.parameters(), *lin2.parameters()]

    def run():
        for p in params:
            p.grad = None
        stages = [_Stage(lin1.weight, lin1.bias, bn, "elu", blocks=[(x, None)]), _Stage(lin2.weight, lin2.bias, None, "identity")]
        y = _autograd.run_stages_autograd(stages, m, True)
        (y * wout).sum().backward()
        return y.detach()[::4097].clone(), [p.grad.clone() for p in params]
    y1, g1 = run()
    try:
        flags.WGRAD_F16X3 = flags.BN_BWD_PLANES = flags.BN_ACT_PLANES = False
        y0, g0 = run()
    finally:
        flags.WGRAD_F16X3 = flags.BN_BWD_PLANES = flags.BN_ACT_PLANES = True
    assert torch.allclose(y1, y0, rtol=1e-5, atol=1e-5)
    for a, b, name in zip(g1, g0, ("W1", "b1", "gamma", "beta", "W2", "b2")):
        if name == "b1":
            continue                        # (in front of a train-mode BatchNorm: the true gradient is zero, both are rounding noise)
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), name

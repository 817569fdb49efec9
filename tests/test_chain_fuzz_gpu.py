"""Randomised shapes through the fused chain entry point (gsn_mlp_chain_fwd_hip via layers._launch_stages / run_stages):
block counts and widths (float4-stageable or not), gathered / direct blocks, one or two stages, fused scatter-add, tails.
Every case is checked against fp64 torch at the 1e-5 parity bar; the bf16x6 and the fp32 kernels are both reachable."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel_err(a, b, floor=1e-30):
    return (a - b).abs().max().item() / max(b.abs().max().item(), floor)


def _case(rng, dev, large=False):
    from gsn_amd.layers import _Stage, _launch_stages, _chain_fits, run_stages, _csr_for
    two = rng.random() < 0.5
    seg = (not two) and rng.random() < 0.6
    mult4 = rng.random() < 0.7
    n_blocks = int(rng.integers(1, 5))
    kmax = 160 if (two or not seg) else 80
    widths = []
    for _ in range(n_blocks):
        w = int(rng.integers(1, 12)) * 4 if mult4 else int(rng.integers(1, 45))
        if sum(widths) + w > kmax:
            break
        widths.append(w)
    if not widths:
        widths = [8]
    n1 = int(rng.choice([5, 16, 32, 33, 64, 100, 128]))
    n2 = int(rng.choice([8, 40, 64, 128]))
    bn = rng.random() < 0.6
    exact = rng.random() < 0.35            # small-integer inputs: exact in bf16 (the kernels skip their zero planes per tile)

    def data(r, w):
        if exact and rng.random() < 0.8:
            return torch.from_numpy(rng.integers(0, 4, (r, w))).to(dev).float()
        return torch.randn(r, w, device=dev)
    act = "relu" if rng.random() < 0.7 else "identity"
    if seg:
        N = int(rng.integers(3, 400)); E = int(rng.integers(1, 3000))
        if large:
            N = int(rng.integers(5000, 30000)); E = int(rng.integers(30000, 90000))
        tgt = torch.from_numpy(rng.integers(0, N, E)).to(dev); src = torch.from_numpy(rng.integers(0, N, E)).to(dev)
        if rng.random() < 0.3:
            tgt[: min(E // 3, 2000)] = int(rng.integers(0, N))            # a long segment (atomics across ranges; an fp32 sum of
                                                                          # more terms than that leaves the 1e-5 bar by rounding alone)
        ei = torch.stack([src, tgt], 0)
        csr = _csr_for(ei, 1, N)
        blocks, cols = [], []
        for i, w in enumerate(widths):
            kind = i % 3
            if kind == 2:
                d = data(E, w); blocks.append((d, csr.perm)); cols.append(d)
            else:
                d = data(N, w); idx = csr.tgt if kind == 0 else csr.src
                blocks.append((d, idx)); cols.append(d[(tgt if kind == 0 else src)])
        K = sum(widths)
        W = torch.randn(n1, K, device=dev) / K ** 0.5; b = torch.randn(n1, device=dev)
        st = _Stage(W, b, None, act, blocks)
        h = torch.cat(cols, 1).double() @ W.double().T + b.double()
        if bn:
            mean, scale, shift = torch.randn(n1, device=dev), torch.rand(n1, device=dev) + 0.5, torch.randn(n1, device=dev)
            st.bn_params = (mean, scale, shift)
            h = (h - mean.double()) * scale.double() + shift.double()
        msg = torch.relu(h) if act == "relu" else h
        ref = torch.zeros(N, n1, dtype=torch.float64, device=dev).index_add_(0, tgt, msg)
        out = run_stages([st], E, False, csr=csr)
        return out, ref, ("seg", widths, n1, E, N)
    M = int(rng.choice([1, 31, 32, 33, 63, 64, 65, 500, 2047, 4100]))
    if large:
        M = int(rng.choice([20000, 33000, 50001, 70000]))
    gathered = rng.random() < 0.4
    blocks, cols = [], []
    for i, w in enumerate(widths):
        if gathered and i % 2 == 0:
            d = data(57, w)
            idx = torch.from_numpy(rng.integers(0, 57, M)).to(dev).to(torch.int32 if i % 4 == 0 else torch.int64)
            blocks.append((d, idx)); cols.append(d[idx.long()])
        else:
            d = data(M, w); blocks.append((d, None)); cols.append(d)
    ref = torch.cat(cols, 1).double()
    stages, k = [], ref.shape[1]
    hidden = [n1, n2] if two else [n1]
    for s, n_out in enumerate(hidden):
        W = torch.randn(n_out, k, device=dev) / k ** 0.5; b = torch.randn(n_out, device=dev)
        st = _Stage(W, b, None, act if (s < len(hidden) - 1 or len(hidden) == 1) else "identity", blocks if s == 0 else ())
        h = ref @ W.double().T + b.double()
        if bn:
            mean, scale, shift = torch.randn(n_out, device=dev), torch.rand(n_out, device=dev) + 0.5, torch.randn(n_out, device=dev)
            st.bn_params = (mean, scale, shift)
            h = (h - mean.double()) * scale.double() + shift.double()
        ref = torch.relu(h) if st.act == "relu" else h
        stages.append(st); k = n_out
    assert _chain_fits(stages)
    return _launch_stages(stages, M), ref, ("chain", widths, hidden, M, gathered)


@pytest.mark.parametrize("seed", range(4))
def test_random_chain_shapes_over_many_row_tiles_vs_fp64(seed):
    """The same random shapes with 20-90 k rows: every persistent workgroup walks several row tiles (the small cases above
    never leave a workgroup's first tile)."""
    rng = np.random.default_rng(7000 + seed)
    dev = torch.device("cuda")
    for _ in range(14):
        out, ref, what = _case(rng, dev, large=True)
        assert out.shape == ref.shape, what
        assert rel_err(out.double(), ref) < TOL, what
        err = (out.double() - ref).abs()
        assert not bool((err > 1e-4 * ref.abs().max()).any()), what


@pytest.mark.parametrize("seed", range(6))
def test_random_chain_shapes_vs_fp64(seed):
    rng = np.random.default_rng(1000 + seed)
    dev = torch.device("cuda")
    for _ in range(40):
        out, ref, what = _case(rng, dev)
        assert out.shape == ref.shape, what
        assert rel_err(out.double(), ref) < TOL, what


@pytest.mark.parametrize("m,widths,n,act,bn", [(1000, [132], 256, "relu", True), (4097, [128, 128, 4], 384, "identity", False),
                                                (130, [300], 600, "relu", False), (777, [64, 200], 130, "tanh", True),
                                                (50000, [128], 256, "relu", True), (3000, [400, 240], 200, "elu", True),
                                                (257, [36], 131, "relu", False), (70000, [64], 1028, "identity", False),
                                                (20000, [352], 300, "relu", False), (60000, [600], 256, "relu", True),
                                                (1, [4], 132, "relu", False), (5, [8, 4], 200, "identity", True), (129, [32], 129, "relu", True)])
def test_direct_rows_on_the_fp16x3_linear_kernel(m, widths, n, act, bn, capfd, monkeypatch):
    """gsn_linear_f16x3_fwd_hip (direct rows, n_out > 128): several input blocks, K not a multiple of the slice and wider than the
    pre-pass keeps in registers, a ragged last row tile and column tile, n_out not a multiple of 4 (4-byte output stores), more
    row tiles than workgroups with an odd number of K slices (the shapes on which a 16-byte store's data register was once
    overwritten behind the store), every epilogue -- against fp64, element-wise, with the product's own condition scale as floor."""
    import os
    from gsn_amd import flags, layers
    monkeypatch.setattr(flags, "LINEAR_F16X3_MIN_TILES", 0)       # (products of few tiles go to the bf16x6 kernel's 32-row twin by default: tests/test_linear_small_gpu.py)
    g = torch.Generator().manual_seed(m + n)
    xs = [torch.randn(m, w, generator=g) * (10.0 ** i) for i, w in enumerate(widths)]
    k = sum(widths)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    w[3] *= 1e-12; w[7] *= 1e9                                  # column scales far apart
    b = torch.randn(n, generator=g)
    bnm = None
    if bn:
        bnm = torch.nn.BatchNorm1d(n)
        bnm.running_mean.copy_(torch.rand(n, generator=g) - 0.5); bnm.running_var.copy_(torch.rand(n, generator=g) + 0.5)
        bnm.weight.data.copy_(torch.rand(n, generator=g) + 0.5); bnm.bias.data.copy_(torch.rand(n, generator=g) - 0.5)
        bnm.eval().cuda()
    st = layers._Stage(w.cuda(), b.cuda(), bnm, act, [(x.cuda(), None) for x in xs])
    os.environ["GSN_CHAIN_TRACE"] = "1"
    try:
        y = layers.run_stages([st], m, False).cpu()
        torch.cuda.synchronize()
    finally:
        os.environ.pop("GSN_CHAIN_TRACE", None)
    assert "linear_f16x3_kernel" in capfd.readouterr().err
    x = torch.cat(xs, 1).double()
    pre = x @ w.double().t() + b.double()
    scale = x.abs() @ w.double().abs().t() + b.double().abs()
    if bn:
        s_ = (bnm.weight.double().cpu() / torch.sqrt(bnm.running_var.double().cpu() + bnm.eps))
        pre = (pre - bnm.running_mean.double().cpu()) * s_ + bnm.bias.double().cpu()
        scale = scale * s_.abs() + bnm.bias.double().cpu().abs() + (bnm.running_mean.double().cpu() * s_).abs()
    ref = {"relu": torch.relu, "identity": lambda t: t, "tanh": torch.tanh, "elu": torch.nn.functional.elu}[act](pre)
    err = (y.double() - ref).abs()
    assert bool((err <= 1e-5 * ref.abs() + 2e-6 * scale).all()), float((err / (1e-5 * ref.abs() + 2e-6 * scale)).max())


@pytest.mark.parametrize("m,widths,n", [(1000, [132], 256), (4097, [128, 128, 4], 384), (130, [300], 600), (105083, [300], 600), (60001, [600], 300),
                                         (257, [36], 132), (1, [4], 132), (129, [32], 260), (70000, [64], 1028)])
def test_train_mode_stage_on_the_fp16x3_kernel_rows_and_statistics(m, widths, n, capfd, monkeypatch):
    """gsn_linear_f16x3_fwd_stats_hip (layers._linear_hip with ``stats``: a train-mode BatchNorm stage, models_misc.py:52-58): the pre-BN
    rows against fp64 element-wise, their fp64 column sums / sums of squares against the fp64 rows' own -- ragged last row tile (the rows
    past M must not be counted) and column tile, several blocks, more row tiles than workgroups, stats ADDED to what the buffer holds --
    and against the bf16x6 kernel's statistics of the same stage."""
    import os
    from gsn_amd import flags, layers
    g = torch.Generator().manual_seed(m + n + 1)
    xs = [torch.randn(m, w, generator=g) * (10.0 ** i) + 0.3 for i, w in enumerate(widths)]
    k = sum(widths)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    w[3] *= 1e-6; w[7] *= 1e4
    b = torch.randn(n, generator=g)
    blocks = [(x.cuda(), None) for x in xs]
    wd, bd = w.cuda(), b.cuda()
    got = {}
    for name, f16 in (("fp16x3", True), ("bf16x6", False)):
        monkeypatch.setattr(flags, "LINEAR_F16X3_STATS", f16)
        monkeypatch.setattr(flags, "LINEAR_F16X3_MIN_TILES", 0)
        stats = torch.full((2, n), 1.5, dtype=torch.float64, device="cuda")
        os.environ["GSN_CHAIN_TRACE"] = "1"
        try:
            y = layers._linear_hip(blocks, wd, bd, None, None, None, 0, m, out=True, stats=stats)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("GSN_CHAIN_TRACE", None)
        assert ("linear_f16x3_kernel" in capfd.readouterr().err) == f16
        got[name] = (y.cpu().double(), stats.cpu() - 1.5)
    x = torch.cat(xs, 1).double()
    ref = x @ w.double().t() + b.double()
    scale = x.abs() @ w.double().abs().t() + b.double().abs()
    for name, (y, st) in got.items():
        err = (y - ref).abs()
        assert bool((err <= 1e-5 * ref.abs() + 2e-6 * scale).all()), (name, float((err / (1e-5 * ref.abs() + 2e-6 * scale)).max()))
        # the statistics are those of the rows the kernel WROTE (fp32 values, fp64 sums): tight; and within the product's error of the fp64 rows'
        assert bool(((st[0] - y.sum(0)).abs() <= 1e-11 * y.abs().sum(0) + 1e-300).all()), name
        assert bool(((st[1] - (y * y).sum(0)).abs() <= 1e-11 * (y * y).sum(0) + 1e-300).all()), name
        assert bool(((st[0] - ref.sum(0)).abs() <= 2e-6 * scale.sum(0)).all()), name
    assert bool(((got["fp16x3"][1][1] - got["bf16x6"][1][1]).abs() <= 1e-4 * got["bf16x6"][1][1].abs() + 1e-300).all())


@pytest.mark.parametrize("m,k,widths", [(70000, 132, [96]), (70000, 128, [64]), (50000, 36, [32]), (70000, 100, [64, 64]), (40000, 160, [128, 96]),
                                          (70000, 260, [64]), (70000, 132, [128])])
def test_narrow_stages_over_many_row_tiles(m, k, widths):
    """Stages with fewer than 128 output columns over more rows than one pass of the persistent workgroups covers (regression: see
    test_narrow_layers_on_a_big_batch_vs_oracle), one and two stages, against fp64 element-wise."""
    from gsn_amd import flags, layers
    g = torch.Generator().manual_seed(m + k + sum(widths))
    x = torch.randn(m, k, generator=g)
    stages, cur, ref = [], k, x.double()
    for i, n in enumerate(widths):
        w = torch.randn(n, cur, generator=g) / cur ** 0.5
        b = torch.randn(n, generator=g)
        stages.append(layers._Stage(w.cuda(), b.cuda(), None, "relu", [(x.cuda(), None)] if i == 0 else []))
        ref = torch.relu(ref @ w.double().t() + b.double())
        cur = n
    y = layers.run_stages(stages, m, False).cpu()
    assert y.shape == ref.shape
    err = (y.double() - ref).abs()
    bad = err > 1e-5 * ref.abs().max()
    assert not bool(bad.any()), "rows %s" % bad.any(1).nonzero().flatten()[:8].tolist()


@pytest.mark.parametrize("seed", range(3))
def test_random_wide_linear_shapes_over_many_row_tiles_vs_fp64(seed):
    """gsn_linear_fwd_hip / gsn_linear_f16x3_fwd_hip (layers._linear_hip: the stages too wide for the chain kernels) with random
    block lists (direct and gathered rows), K 164-700, n_out 40-600, 20-70 k rows, with and without the train-mode column sums:
    output and fp64 column sums / sums of squares against fp64."""
    from gsn_amd import flags, layers
    rng = np.random.default_rng(9000 + seed)
    dev = torch.device("cuda")
    for _ in range(8):
        M = int(rng.choice([20000, 33001, 50001, 70000]))
        n_blocks = int(rng.integers(1, 5))
        K = int(rng.integers(41, 176)) * 4
        cuts = sorted(set(int(c) * 4 for c in rng.integers(1, K // 4, n_blocks - 1))) if n_blocks > 1 else []
        widths = [b - a for a, b in zip([0] + cuts, cuts + [K])]
        N = int(rng.choice([40, 128, 132, 300, 600]))
        gathered = rng.random() < 0.5
        blocks, cols = [], []
        for i, w in enumerate(widths):
            if gathered and i % 2 == 0:
                d = torch.randn(997, w, device=dev)
                idx = torch.from_numpy(rng.integers(0, 997, M)).to(dev).to(torch.int32 if i % 4 == 0 else torch.int64)
                blocks.append((d, idx)); cols.append(d[idx.long()])
            else:
                d = torch.randn(M, w, device=dev); blocks.append((d, None)); cols.append(d)
        x = torch.cat(cols, 1).double()
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        act = int(rng.integers(0, 2))
        want_stats = rng.random() < 0.4
        what = (M, widths, N, gathered, act, want_stats)
        if want_stats:                                  # train-mode first pass: pre-activation rows + their column sums
            stats = torch.zeros(2 * N, dtype=torch.float64, device=dev)
            y = layers._linear_hip(blocks, W, b, None, None, None, 0, M, stats=stats)
            ref = x @ W.double().T + b.double()
            s_ref = torch.cat([ref.sum(0), (ref * ref).sum(0)])
            # (the kernel sums its rounded fp32 outputs in fp64: against the exact sums the error is a random walk of M roundings,
            #  to be measured against the column's mass, not against a first moment that may cancel to ~0)
            mass = torch.cat([ref.abs().sum(0), (ref * ref).sum(0)])
            assert float(((stats - s_ref).abs() / mass).max()) < 1e-6, what
        else:
            y = layers._linear_hip(blocks, W, b, None, None, None, act, M)
            ref = x @ W.double().T + b.double()
            if act == 1:
                ref = torch.relu(ref)
        err = (y.double() - ref).abs()
        assert not bool((torch.isnan(err) | (err > 1e-5 * ref.abs().max())).any()), what

"""The 32-row-tile twin of the bf16x6 dense kernel (csrc/linear.hip: linear_fwd_bf16_small_kernel), taken when the product has few
128-row tiles (the dense stages of a training step at the reference's batch sizes, train_test_funcs.py:88-106): every staging path
(float4 / scalar / a transposed weight view), gathered blocks with 32- and 64-bit indices, bias + BatchNorm vectors + activation,
column statistics, ragged row counts and output widths, against float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(blocks, w, bias, bn, act, m):
    cols = []
    for d, idx in blocks:
        d = d.double()
        cols.append(d[:m] if idx is None else d[idx.long()])
    y = torch.cat(cols, 1) @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    pre = y
    if bn is not None:
        mean, scale, shift = (v.double() for v in bn)
        y = (y - mean) * scale + shift
    y = {0: lambda t: t, 1: torch.relu, 2: torch.nn.functional.elu, 3: torch.tanh}[act](y)
    return y, pre


CASES = [  # m_rows, block widths, gathered?, n_out, act, bn, bias, transposed weight view
    (837, [300], [0], 600, 1, True, True, False),
    (5924, [128, 128, 12, 4], [32, 64, 0, 0], 128, 1, False, True, False),
    (2903, [128, 128, 1], [0, 0, 0], 128, 0, False, True, False),
    (33, [7, 5], [64, 0], 130, 2, True, False, False),
    (1, [4], [0], 1, 3, False, True, False),
    (1000, [64], [0], 300, 0, False, False, True),
    (3071, [40, 24, 8, 8, 16], [32, 32, 0, 64, 0], 257, 1, True, True, False),
]


@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) + "x" + str(sum(c[1])) + "x" + str(c[3]) for c in CASES])
@pytest.mark.parametrize("with_stats", [False, True])
def test_small_row_counts_against_float64(case, with_stats, monkeypatch):
    from gsn_amd import flags, layers
    m, widths, gath, n_out, act, use_bn, use_bias, transposed = case
    monkeypatch.setattr(flags, "LINEAR_F16X3", False)
    rng = np.random.default_rng(m * 31 + n_out)
    dev = torch.device("cuda")
    blocks = []
    for wd, g in zip(widths, gath):
        if g:
            src_rows = int(rng.integers(5, 400))
            d = torch.from_numpy(rng.standard_normal((src_rows, wd)).astype(np.float32)).to(dev)
            idx = torch.from_numpy(rng.integers(0, src_rows, m)).to(dev)
            blocks.append((d, idx.to(torch.int32) if g == 32 else idx))
        else:
            blocks.append((torch.from_numpy(rng.standard_normal((m, wd)).astype(np.float32)).to(dev), None))
    k = sum(widths)
    w = torch.from_numpy(rng.standard_normal((n_out, k)).astype(np.float32) / np.sqrt(k)).to(dev)
    if transposed:
        w = w.t().contiguous().t()          # same values, row stride 1
        assert not w.is_contiguous()
    bias = torch.from_numpy(rng.standard_normal(n_out).astype(np.float32)).to(dev) if use_bias else None
    bn = tuple(torch.from_numpy(v.astype(np.float32)).to(dev) for v in (rng.standard_normal(n_out), rng.uniform(0.5, 2.0, n_out), rng.standard_normal(n_out))) \
        if (use_bn and not with_stats) else None
    if with_stats:          # a train-mode stage: raw pre-BN rows + their column sums / sums of squares
        stats = torch.zeros(2 * n_out, dtype=torch.float64, device=dev)
        y = layers._linear_hip(blocks, w, bias, None, None, None, 0, m, out=True, stats=stats)
        ref, _ = _ref(blocks, w, bias, None, 0, m)
        scale = float(ref.abs().max())
        assert float((y.double() - ref).abs().max()) <= 2e-6 * scale
        assert float((stats[:n_out] - y.double().sum(0)).abs().max()) <= 1e-9 * scale * m
        assert float((stats[n_out:] - (y.double() ** 2).sum(0)).abs().max()) <= 1e-9 * scale * scale * m
        return
    y = layers._linear_hip(blocks, w, bias, *(bn if bn else (None, None, None)), act, m)
    ref, _ = _ref(blocks, w, bias, bn, act, m)
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())


def test_a_row_does_not_depend_on_the_tile_height():
    """The same rows as part of a product with many 128-row tiles (the 128-row kernel) and as a small product (32-row tiles): identical bits."""
    from gsn_amd import flags, layers
    old = flags.LINEAR_F16X3
    flags.LINEAR_F16X3 = False
    try:
        g = torch.Generator().manual_seed(8)
        big = torch.randn(40000, 272, generator=g).cuda()
        w = (torch.randn(128, 272, generator=g) / 16).cuda()
        b = torch.randn(128, generator=g).cuda()
        y_big = layers._linear_hip([(big, None)], w, b, None, None, None, 1, big.shape[0])
        y_small = layers._linear_hip([(big[:3000].contiguous(), None)], w, b, None, None, None, 1, 3000)
        assert torch.equal(y_big[:3000], y_small)
    finally:
        flags.LINEAR_F16X3 = old


SPLITK_CASES = [  # m_rows, block widths, gathered?, n_out, bias, transposed weight view
    (837, [600], [0], 300, False, True),        # the input-gradient product of a d = 300 stage at molhiv's batch of 32 graphs
    (837, [300], [0], 600, False, True),
    (837, [300], [0], 600, True, False),        # float4 staging
    (1201, [301], [0], 130, True, False),       # scalar staging, ragged everything
    (2000, [128, 128, 32], [32, 64, 0], 128, True, False),
    (65, [1000], [0], 17, False, True),
]


@pytest.mark.parametrize("case", SPLITK_CASES, ids=[str(c[0]) + "x" + str(sum(c[1])) + "x" + str(c[3]) for c in SPLITK_CASES])
def test_split_k_products_against_float64_and_the_one_range_kernel(case, monkeypatch):
    """gsn_linear_fwd_splitk_hip (K ranges of an output tile on several workgroups, added into a zero-filled output) against float64 and
    against the same product with one workgroup per tile."""
    from gsn_amd import _abi, flags, layers
    m, widths, gath, n_out, use_bias, transposed = case
    monkeypatch.setattr(flags, "LINEAR_F16X3", False)
    rng = np.random.default_rng(m * 17 + n_out)
    dev = torch.device("cuda")
    blocks = []
    for wd, g in zip(widths, gath):
        if g:
            src_rows = int(rng.integers(5, 400))
            d = torch.from_numpy(rng.standard_normal((src_rows, wd)).astype(np.float32)).to(dev)
            idx = torch.from_numpy(rng.integers(0, src_rows, m)).to(dev)
            blocks.append((d, idx.to(torch.int32) if g == 32 else idx))
        else:
            blocks.append((torch.from_numpy(rng.standard_normal((m, wd)).astype(np.float32)).to(dev), None))
    k = sum(widths)
    w = torch.from_numpy(rng.standard_normal((n_out, k)).astype(np.float32) / np.sqrt(k)).to(dev)
    if transposed:
        w = w.t().contiguous().t()
    bias = torch.from_numpy(rng.standard_normal(n_out).astype(np.float32)).to(dev) if use_bias else None
    assert _abi.lib().gsn_linear_splitk_plan(m, k, n_out) > 1
    flags.KERNEL_TIMER = None
    y = layers._linear_hip(blocks, w, bias, None, None, None, 0, m, split_k=True)
    y1 = layers._linear_hip(blocks, w, bias, None, None, None, 0, m)          # (not asked for: one workgroup per tile)
    ref, _ = _ref(blocks, w, bias, None, 0, m)
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 3e-6 * scale
    assert float((y - y1).abs().max()) <= 1e-6 * scale          # (the ranges' partial sums are added in fp32: an ulp or two of the largest term)
    # rows past the last tile's end and columns past n_out: nothing written outside [m][n_out] (the output is exactly that large) -- and
    # a second call must not see the first call's sums (fresh zeros from the arena)
    y2 = layers._linear_hip(blocks, w, bias, None, None, None, 0, m, split_k=True)
    assert float((y2.double() - ref).abs().max()) <= 3e-6 * scale
    y3 = layers._linear_hip(blocks, w, bias, None, None, None, 0, m)
    assert torch.equal(y3, y1)                                                 # forward products: the same bits every time


def test_split_k_plan_leaves_activations_statistics_and_large_products_alone():
    from gsn_amd import _abi
    lib = _abi.lib()
    assert lib.gsn_linear_splitk_plan(837, 600, 300) == 4
    assert lib.gsn_linear_splitk_plan(837, 64, 300) == 1            # two slices: nothing to split
    assert lib.gsn_linear_splitk_plan(105083, 600, 300) == 1        # config-4 sizes: the 128-row kernels
    assert lib.gsn_linear_splitk_plan(0, 600, 300) == 1

"""The pipelined propagate kernels of round 5 (csrc/propagate.hip: relu_sum3_kernel with two and three streams and the layer's own term,
cat_pipe_kernel for long segments and the x_j-only gin aggregation, the relu-sum adjoints with every chunk in flight, the folded own-term
adjoint) against plain PyTorch on graphs that exercise every arm of the pipeline: targets without edges, segments of exactly four edges (the
prefetched ids), longer ones (the tail loops), one hub holding a third of all edges, a last partial wave.  Replaces
torch.sparse.sum / index_select of GSN_sparse.py:125-143, GSN_edge_sparse_ogb.py:103-106."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _graph(n, kind, seed):
    """edge_index [2, E] (row 0 = source, row 1 = target) with a chosen in-degree profile."""
    rng = np.random.default_rng(seed)
    if kind == "molecule":            # degrees 0..4
        deg = rng.integers(0, 5, n)
    elif kind == "four":              # exactly the prefetch depth
        deg = np.full(n, 4)
    elif kind == "long":              # readout-like
        deg = rng.integers(5, 40, n)
    else:                             # "hub"
        deg = rng.integers(0, 4, n)
        deg[n // 2] = max(1, int(deg.sum()) // 2)
    tgt = np.repeat(np.arange(n), deg)
    src = rng.integers(0, n, tgt.size)
    perm = rng.permutation(tgt.size)                 # edge order unrelated to the targets
    return torch.from_numpy(np.stack([src[perm], tgt[perm]])).to("cuda")


def _relu_sum_ref(ei, n, a, b, c, self_x=None, eps=None):
    m = a[ei[0]]
    if b is not None:
        m = m + b
    if c is not None:
        m = m + c
    out = torch.zeros(n, a.shape[1], device=a.device, dtype=a.dtype).index_add_(0, ei[1], torch.relu(m))
    if self_x is not None:
        out = out + (1 + (eps if eps is not None else 0)) * self_x
    return out


@pytest.mark.parametrize("kind", ["molecule", "four", "long", "hub"])
@pytest.mark.parametrize("d", [132, 300, 320, 384])
@pytest.mark.parametrize("streams", [2, 3])
def test_relu_sum_forward_and_adjoints_vs_torch(kind, d, streams):
    from gsn_amd.layers import propagate
    n = 1237
    ei = _graph(n, kind, 3)
    E = ei.shape[1]
    g = torch.Generator(device="cpu").manual_seed(d + streams)
    mk = lambda r: torch.randn(r, d, generator=g, dtype=torch.float64).cuda()
    a64, b64, c64 = mk(n), mk(E), (mk(E) if streams == 3 else None)
    eps64 = torch.tensor([0.3], dtype=torch.float64, device="cuda")
    up = mk(n)
    for with_self in (False, True):
        leaves64 = [t.clone().requires_grad_() for t in (a64, b64) + ((c64,) if streams == 3 else ()) + ((eps64,) if with_self else ())]
        a_, b_ = leaves64[0], leaves64[1]
        c_ = leaves64[2] if streams == 3 else None
        e_ = leaves64[-1] if with_self else None
        ref = _relu_sum_ref(ei, n, a_, b_, c_, a_ if with_self else None, e_)
        (ref * up).sum().backward()
        leaves32 = [t.detach().float().requires_grad_() for t in leaves64]
        a, b = leaves32[0], leaves32[1]
        c = leaves32[2] if streams == 3 else None
        e = leaves32[-1] if with_self else None
        out = propagate(1, ei, 1, n, a=a, b=b, c=c, selfs=(a,) if with_self else (), eps=e)
        scale = ref.abs().max().item()
        assert (out.double() - ref).abs().max().item() <= TOL * scale
        (out * up.float()).sum().backward()
        # (fp64 reference: a pre-activation within fp32 rounding of zero may take the other branch of the ReLU in fp32 -- its gradient
        #  element then differs by that element's upstream value; at most a handful among ~10^6, bounded here by count)
        for l32, l64 in zip(leaves32, leaves64):
            diff = (l32.grad.double() - l64.grad).abs()
            bar = 2e-5 * l64.grad.abs().max().item()
            assert int((diff > bar).sum().item()) <= 8, (kind, d, streams, with_self, int((diff > bar).sum().item()))


@pytest.mark.parametrize("kind", ["long", "hub", "molecule"])
@pytest.mark.parametrize("widths", [(128, 0, 0), (0, 300, 0), (0, 128, 0), (0, 40, 0), (64, 40, 8), (0, 600, 0)])
def test_concatenation_and_readout_kernels_vs_torch(kind, widths):
    from gsn_amd.layers import propagate
    da, db, dc = widths
    n = 900
    ei = _graph(n, kind, 5)
    E = ei.shape[1]
    g = torch.Generator(device="cpu").manual_seed(da + db + dc)
    a = torch.randn(n, da, generator=g).cuda() if da else None
    b = torch.randn(E, db, generator=g).cuda() if db else None
    c = torch.randn(E, dc, generator=g).cuda() if dc else None
    out = propagate(0, ei, 1, n, a=a, b=b, c=c)
    parts = ([a[ei[0]]] if da else []) + ([b] if db else []) + ([c] if dc else [])
    ref = torch.zeros(n, da + db + dc, dtype=torch.float64, device="cuda").index_add_(0, ei[1], torch.cat(parts, 1).double())
    # (sums of up to ~10^4 rows on the hub: the bar is relative to the row's own magnitude scale)
    tol = TOL * ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-30) + 1e-30
    assert bool(((out.double() - ref).abs() <= tol + TOL * ref.abs()).all())
    # the same bits as the plain kernel (the pipeline changes no order of additions): forced through GSN_PROP_CP
    import os
    old = os.environ.get("GSN_PROP_CP")
    os.environ["GSN_PROP_CP"] = "0"
    try:
        plain = propagate(0, ei, 1, n, a=a, b=b, c=c)
    finally:
        if old is None:
            del os.environ["GSN_PROP_CP"]
        else:
            os.environ["GSN_PROP_CP"] = old
    assert torch.equal(out, plain)


def test_relu_sum_pipeline_leaves_the_plain_kernels_bits():
    import os
    from gsn_amd.layers import propagate
    n = 5000
    ei = _graph(n, "molecule", 9)
    E = ei.shape[1]
    g = torch.Generator(device="cpu").manual_seed(1)
    a, b, c = torch.randn(n, 300, generator=g).cuda(), torch.randn(E, 300, generator=g).cuda(), torch.randn(E, 300, generator=g).cuda()
    eps = torch.tensor([0.25], device="cuda")
    outs = {}
    for rs in ("0", "16,1,0", "16,2,0", "32,2,0", "32,4,0", "64,2,0", "32,2,1"):
        os.environ["GSN_PROP_RS"] = rs
        try:
            outs[rs] = (propagate(1, ei, 1, n, a=a, b=b, c=c), propagate(1, ei, 1, n, a=a, b=b, c=None, selfs=(a,), eps=eps))
        finally:
            del os.environ["GSN_PROP_RS"]
    for rs, (y3, y2) in outs.items():
        assert torch.equal(y3, outs["0"][0]) and torch.equal(y2, outs["0"][1]), rs

"""gsn_fold_weights_{fwd,bwd}_hip: update_fn's first weight with msg_fn's last Linear folded in (GSN_edge_sparse.py:153-170 evaluated as
cat(x, S, deg) [W3x | W3a W2 | W3a b2]^T), forward and adjoint against float64 tensor algebra; the `general` training layer gives the same
gradients with the kernel and with the composition of dense stages it replaces."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,d_x,A,H", [(128, 128, 128, 128), (64, 28, 96, 40), (1, 0, 1, 1), (300, 300, 300, 600), (17, 5, 259, 3), (130, 64, 1, 257)])
def test_fold_forward_and_adjoint_against_float64(R, d_x, A, H):
    from gsn_amd.layers import _FoldWeightsFn
    g = torch.Generator().manual_seed(R * 7 + A)
    w3 = torch.randn(R, d_x + A, generator=g).cuda().requires_grad_()
    w2 = torch.randn(A, H, generator=g).cuda().requires_grad_()
    b2 = torch.randn(A, generator=g).cuda().requires_grad_()
    out = _FoldWeightsFn.apply(w3, w2, b2, d_x)
    assert out.shape == (R, d_x + H + 1)
    gy = torch.randn(out.shape, generator=g).cuda()
    got = torch.autograd.grad(out, [w3, w2, b2], gy)
    w3d, w2d, b2d = (t.detach().double().requires_grad_() for t in (w3, w2, b2))
    ref = torch.cat([w3d[:, :d_x], w3d[:, d_x:] @ w2d, (w3d[:, d_x:] @ b2d).unsqueeze(1)], 1)
    want = torch.autograd.grad(ref, [w3d, w2d, b2d], gy.double())
    assert torch.equal(out[:, :d_x], w3[:, :d_x])
    scale = float(ref.detach().abs().max())
    assert float((out.double() - ref.detach()).abs().max()) <= 2e-6 * scale
    for a, b in zip(got, want):
        assert a.shape == b.shape
        assert float((a.double() - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-30


def test_fold_takes_a_row_strided_gradient():
    from gsn_amd.layers import _FoldWeightsFn
    g = torch.Generator().manual_seed(3)
    R, d_x, A, H = 40, 12, 33, 21
    w3 = torch.randn(R, d_x + A, generator=g).cuda().requires_grad_()
    w2 = torch.randn(A, H, generator=g).cuda().requires_grad_()
    b2 = torch.randn(A, generator=g).cuda().requires_grad_()
    out = _FoldWeightsFn.apply(w3, w2, b2, d_x)
    wide = torch.randn(R, d_x + H + 9, generator=g).cuda()
    a = torch.autograd.grad(out, [w3, w2, b2], wide[:, :d_x + H + 1], retain_graph=True)
    b = torch.autograd.grad(out, [w3, w2, b2], wide[:, :d_x + H + 1].contiguous())
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("cls", ["GSN_edge_sparse", "GSN_sparse"])
def test_general_training_layer_same_gradients_with_and_without_the_fold_kernel(cls, monkeypatch):
    from gsn_amd import flags, layers
    torch.manual_seed(11)
    n, E, d = 300, 1400, 64
    x = torch.randn(n, 24).cuda()
    ei = torch.randint(0, n, (2, E)).cuda()
    deg = torch.zeros(n, device="cuda")
    base = dict(d_in=24, d_id=6, d_degree=1, degree_as_tag=False, retain_features=True, d_msg=d, d_up=d, d_h=[d], seed=0, activation_name="relu",
                bn=True, msg_kind="general", flow="source_to_target")
    if cls == "GSN_edge_sparse":
        layer = layers.GSN_edge_sparse(d_ef=4, id_scope="local", **base).cuda()
        kw = dict(identifiers=torch.randn(E, 6).cuda(), degrees=deg, edge_features=torch.randn(E, 4).cuda())
    else:
        layer = layers.GSN_sparse(id_scope="global", **base).cuda()
        kw = dict(identifiers=torch.randn(n, 6).cuda(), degrees=deg)
    layer.train()
    outs = {}
    for on in (True, False):
        monkeypatch.setattr(flags, "FOLD_KERNEL", on)
        layer.zero_grad(set_to_none=True)
        xx = x.clone().requires_grad_()
        y = layer(xx, ei, **kw)
        (y * torch.linspace(-1, 1, y.numel(), device=y.device).view_as(y)).sum().backward()
        outs[on] = [y.detach().clone(), xx.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
    assert len(outs[True]) == len(outs[False]) > 4
    whole = max(float(b.abs().max()) for b in outs[False][1:])
    for a, b in zip(outs[True], outs[False]):       # (a bias in front of a train-mode BatchNorm has the gradient 0: rounding noise on both sides)
        den = max(float(b.abs().max()), 1e-3 * whole)
        assert float((a - b).abs().max()) <= 2e-4 * den

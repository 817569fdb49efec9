"""The adjoint of the ogb layers' own term (1 + eps) x inside the node pass of the propagate adjoint (gsn_propagate_bwd_fold_self_hip,
flags.FOLD_SELF_ADJOINT; GSN_edge_sparse_ogb.py:63-84: out = (1 + eps) x + sum relu(x_j + id_e + e_e)) against the two-function path
(gsn_propagate_pad_bwd_hip + gsn_propagate_self_bwd_hip + autograd's sum of the two gradients of x): the same gradients for x, eps, the per-edge
inputs and the layer's parameters."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grads(fold, cls_name, d, train, n_graphs, with_ef):
    from gsn_amd import flags, layers, synth
    dev = torch.device("cuda", 0)
    old = flags.FOLD_SELF_ADJOINT
    flags.FOLD_SELF_ADJOINT = fold
    try:
        b = synth.zinc_shape_batch(n_graphs, seed=5)
        N, E = b.num_nodes, b.num_edges
        ei = torch.from_numpy(b.edge_index).to(dev)
        g = torch.Generator(device="cpu").manual_seed(d + n_graphs)
        x = torch.randn(N, d, generator=g).to(dev).requires_grad_()
        ids = torch.randn(E, d, generator=g).to(dev).requires_grad_()
        ef = torch.randn(E, d, generator=g).to(dev).requires_grad_() if with_ef else None
        up = torch.randn(N, d, generator=g).to(dev)
        kw = dict(d_in=d, d_ef=d, d_degree=1, degree_as_tag=False, retain_features=True, d_msg=d, d_up=d, d_h=[2 * d], seed=0,
                  activation_name="elu", bn=True, train_eps=True, flow="source_to_target")
        torch.manual_seed(11)                  # (the layers draw their initial weights from the global generator)
        if cls_name == "GSN_edge_sparse_ogb":
            lay = layers.GSN_edge_sparse_ogb(d_id=d, id_scope="local", **kw)
        else:
            lay = layers.MPNN_edge_sparse_ogb(**kw)
        lay = lay.to(dev).train(train)
        with torch.no_grad():
            lay.eps.fill_(0.37)
        if cls_name == "GSN_edge_sparse_ogb":
            y = lay(x, ei, degrees=None, identifiers=ids, edge_features=ef)
        else:
            y = lay(x, ei, degrees=None, identifiers=None, edge_features=ids)
        (y * up).sum().backward()
        out = {"x": x.grad.clone(), "ids": ids.grad.clone(), "eps": lay.eps.grad.clone()}
        if ef is not None:
            out["ef"] = ef.grad.clone()
        for n, p in lay.named_parameters():
            if p.grad is not None and n != "eps":
                out["p." + n] = p.grad.clone()
        return out
    finally:
        flags.FOLD_SELF_ADJOINT = old


@pytest.mark.parametrize("cls_name,with_ef", [("GSN_edge_sparse_ogb", True), ("GSN_edge_sparse_ogb", False), ("MPNN_edge_sparse_ogb", False)])
@pytest.mark.parametrize("d,n_graphs", [(300, 200), (136, 30), (320, 3)])
@pytest.mark.parametrize("train", [True, False])
def test_folded_self_adjoint_equals_the_two_function_path(cls_name, with_ef, d, n_graphs, train):
    if cls_name == "GSN_edge_sparse_ogb" and not with_ef:
        pytest.skip("the layer takes edge_features=None only from the fused encoders' caller; covered by tests/test_fused_encoders_gpu.py")
    g0 = _grads(False, cls_name, d, train, n_graphs, with_ef)
    g1 = _grads(True, cls_name, d, train, n_graphs, with_ef)
    assert set(g0) == set(g1)
    scale = max(v.abs().max().item() for v in g0.values())
    for k in g0:
        s = g0[k].abs().max().item()
        # (the message relu sees the same forward bits on both sides; what differs is one fused multiply-add per element of g_x and the order
        #  of the atomics in the dense adjoints)
        assert (g1[k] - g0[k]).abs().max().item() <= 2e-5 * s + 1e-6 * scale, (k, (g1[k] - g0[k]).abs().max().item(), s)


def test_the_folded_pass_is_what_runs_and_where_it_cannot_the_two_functions_do(capfd):
    from gsn_amd import _abi
    lib = _abi.lib()
    dev = torch.device("cuda", 0)
    # no edges: nothing to fold into -- GSN_E_UNSUPPORTED, nothing launched
    a = torch.randn(4, 8, device=dev); go = torch.randn(4, 8, device=dev); ga = torch.empty(4, 8, device=dev)
    seg = torch.zeros(5, dtype=torch.int32, device=dev)
    rc = lib.gsn_propagate_bwd_fold_self_hip(4, 0, None, None, seg.data_ptr(), None, a.data_ptr(), 8, None, 0, None, 0, go.data_ptr(), ga.data_ptr(),
                                             None, None, None, None, _abi.current_stream())
    assert rc == -2

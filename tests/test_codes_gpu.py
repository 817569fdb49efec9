"""Layers fed integer codes (layers.Codes) instead of the dense one-hot floats: the weight-row-gather edge stage
(gsn_code_stage_fwd_hip) must reproduce the dense formulation, which itself is pinned to the reference by
tests/test_layers_gpu.py.  Tolerance: 1e-5 relative (fp32, different summation order of the few non-zero terms)."""
import numpy as np
import pytest
import torch

from gsn_amd import layers, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _graph(n_graphs, seed, isolated=True):
    b = synth.zinc_shape_batch(n_graphs, seed=seed)
    ei = torch.from_numpy(b.edge_index).to(DEV)
    n = int(b.node_ptr[-1]) + (3 if isolated else 0)       # trailing vertices without edges
    return ei, n


def _codes(rng, rows, n_classes):
    return torch.from_numpy(rng.integers(0, n_classes, size=(rows, len(n_classes)))).to(DEV)


def _close(a, b, tol=1e-5):
    scale = max(float(b.abs().max()), 1e-6)
    return float((a - b).abs().max()) <= tol * scale * 4


CASES = [
    # cls, scope, x classes, id classes, ef classes, d, act, bn, flow
    ("GSN_edge_sparse", "local", [28], [3, 4, 4, 5], [4], 128, "relu", True, "source_to_target"),
    ("GSN_edge_sparse", "global", [28], [3, 4, 4, 5, 2, 2], [4], 128, "relu", True, "source_to_target"),   # 15 slots
    ("GSN_edge_sparse", "local", [9, 3], [7], [4, 2], 64, "elu", False, "target_to_source"),
    ("GSN_edge_sparse", "local", [28], [3, 4], [4], 200, "tanh", True, "source_to_target"),
    ("GSN_sparse", "local", [5], [3, 29, 34], None, 64, "relu", True, "source_to_target"),
    ("GSN_sparse", "global", [5], [6, 2], None, 96, "identity", False, "source_to_target"),
    ("MPNN_edge_sparse", None, [28], None, [4], 128, "relu", True, "source_to_target"),
    ("MPNN_sparse", None, [17], None, None, 32, "relu", False, "source_to_target"),
]


def _make(case, seed):
    cls, scope, xc, ic, ec, d, act, bn, flow = case
    torch.manual_seed(seed)
    kw = dict(d_in=sum(xc), d_degree=1, degree_as_tag=False, retain_features=True, d_msg=d, d_up=d, d_h=[d], seed=seed,
              activation_name=act, bn=bn, msg_kind="general", flow=flow)
    if ic is not None:
        kw.update(d_id=sum(ic), id_scope=scope)
    if ec is not None:
        kw.update(d_ef=sum(ec))
    return getattr(layers, cls)(**kw).to(DEV)


@pytest.mark.parametrize("train", [False, True], ids=["eval", "train"])
@pytest.mark.parametrize("case", CASES, ids=["%s-%s-%d-%s" % (c[0], c[1], c[5], c[6]) for c in CASES])
def test_codes_equal_dense(case, train):
    _codes_equal_dense(case, train, 40)


@pytest.mark.parametrize("case", CASES, ids=["%s-%s-%d-%s" % (c[0], c[1], c[5], c[6]) for c in CASES])
def test_codes_equal_dense_on_many_row_tiles(case):
    """The same on 2048 graphs (97 k edge rows: several row tiles per persistent workgroup of the weight-row-gather stage)."""
    _codes_equal_dense(case, False, 2048)


def _codes_equal_dense(case, train, n_graphs):
    cls, scope, xc, ic, ec, d, act, bn, flow = case
    rng = np.random.default_rng(3)
    ei, n = _graph(n_graphs, 5)
    E = ei.shape[1]
    layer = _make(case, 1)
    layer.train(train)
    xcodes = layers.Codes(_codes(rng, n, xc), xc)
    kw_c, kw_d = {"degrees": None}, {"degrees": None}
    if ic is not None:
        idc = layers.Codes(_codes(rng, E if scope == "local" else n, ic), ic)
        kw_c["identifiers"], kw_d["identifiers"] = idc, idc.dense().clone()
    if ec is not None:
        efc = layers.Codes(_codes(rng, E, ec), ec)
        kw_c["edge_features"], kw_d["edge_features"] = efc, efc.dense().clone()
    calls = []
    orig = layers._code_stage_segsum

    def spy(*a, **k):
        r = orig(*a, **k)
        calls.append(r is not None)
        return r

    packed = []
    orig_p = layers._SparseLayer._fused_on_code_packs

    def spy_p(self, *a, **k):
        r = orig_p(self, *a, **k)
        packed.append(r is not None)
        return r

    layers._code_stage_segsum = spy
    layers._SparseLayer._fused_on_code_packs = spy_p
    try:
        state = {k: v.clone() for k, v in layer.state_dict().items()}
        with torch.no_grad():
            y_c = layer(xcodes, ei, **kw_c)
        state_c = {k: v.clone() for k, v in layer.state_dict().items()}
        layer.load_state_dict(state)
        n_calls = len(calls)
        with torch.no_grad():
            y_d = layer(xcodes.dense().clone(), ei, **kw_d)
        assert len(calls) == n_calls, "dense inputs must not take the code path"
    finally:
        layers._code_stage_segsum = orig
        layers._SparseLayer._fused_on_code_packs = orig_p
    # (eval-mode shapes that fit the exact fp16 row packs run the packed-row kernel on packs made straight from the codes; the others
    #  the weight-row-gather stage -- either way no dense one-hot)
    assert calls == [True] or (packed and packed[-1] and not calls), "neither the code-gather stage nor the packed-row kernel ran"
    assert y_c.shape == (n, d)
    assert _close(y_c, y_d), float((y_c - y_d).abs().max())
    for k, v in layer.state_dict().items():       # BatchNorm running statistics advance identically
        assert torch.allclose(v.float(), state_c[k].float(), rtol=1e-5, atol=1e-6), k


def test_codes_backward_matches_dense():
    case = CASES[0]
    rng = np.random.default_rng(9)
    ei, n = _graph(12, 2)
    E = ei.shape[1]
    layer = _make(case, 4).train()
    xc, idc, efc = (layers.Codes(_codes(rng, r, c), c) for r, c in ((n, case[2]), (E, case[3]), (E, case[4])))
    gy = torch.randn(n, case[5], device=DEV)
    grads = []
    for inputs in ((xc, idc, efc), (xc.dense().clone(), idc.dense().clone(), efc.dense().clone())):
        layer.zero_grad()
        y = layer(inputs[0], ei, identifiers=inputs[1], edge_features=inputs[2], degrees=None)
        (y * gy).sum().backward()
        grads.append({k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None})
    assert set(grads[0]) == set(grads[1]) and len(grads[0]) >= 8
    for k in grads[0]:
        scale = max(float(grads[1][k].abs().max()), 1e-3)
        assert float((grads[0][k] - grads[1][k]).abs().max()) <= 2e-4 * scale, k


def test_mixed_inputs_fall_back_to_dense_stage():
    case = CASES[0]
    rng = np.random.default_rng(1)
    ei, n = _graph(8, 3)
    E = ei.shape[1]
    layer = _make(case, 2).eval()
    xc, idc, efc = (layers.Codes(_codes(rng, r, c), c) for r, c in ((n, case[2]), (E, case[3]), (E, case[4])))
    with torch.no_grad():
        y_all = layer(xc, ei, identifiers=idc, edge_features=efc, degrees=None)
        y_mix = layer(xc.dense(), ei, identifiers=idc, edge_features=efc, degrees=None)     # float x, coded ids / bonds
        y_gin = None
    assert _close(y_mix, y_all)


def test_codes_on_gin_and_ogb_layers_densify():
    rng = np.random.default_rng(2)
    ei, n = _graph(6, 4, isolated=False)
    E = ei.shape[1]
    torch.manual_seed(0)
    gin = layers.GSN_edge_sparse(d_in=28, d_ef=4, d_id=7, d_degree=1, degree_as_tag=False, retain_features=True,
                                 id_scope="local", d_msg=32, d_up=32, d_h=[32], seed=0, activation_name="relu", bn=False,
                                 msg_kind="gin", flow="source_to_target", id_embedding="one_hot_encoder", extend_dims=True,
                                 edge_embedding="one_hot_encoder").to(DEV).eval()
    xc, idc, efc = layers.Codes(_codes(rng, n, [28]), [28]), layers.Codes(_codes(rng, E, [7]), [7]), layers.Codes(_codes(rng, E, [4]), [4])
    with torch.no_grad():
        a = gin(xc, ei, identifiers=idc, edge_features=efc, degrees=None)
        b = gin(xc.dense(), ei, identifiers=idc.dense(), edge_features=efc.dense(), degrees=None)
    assert torch.equal(a, b)


def test_out_of_range_code_raises():
    case = CASES[6]
    rng = np.random.default_rng(0)
    ei, n = _graph(4, 1)
    layer = _make(case, 0).eval()
    bad = _codes(rng, ei.shape[1], [4])
    bad[5, 0] = 4
    with pytest.raises(IndexError):
        with torch.no_grad():
            layer(layers.Codes(_codes(rng, n, [28]), [28]), ei, edge_features=layers.Codes(bad, [4]), degrees=None)

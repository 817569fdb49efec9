"""Encoders vs outputs of the reference's utils_encoding.encode and utils_graph_learning.DiscreteEmbedding
(tests/golden/dataset.npz)."""
import os

import numpy as np
import pytest
import torch

from gsn_amd import data as gdata
from gsn_amd import encoding

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "dataset.npz"), allow_pickle=False)


def _source_graphs(gold, key):
    out = []
    for g in range(int(gold[key + "/n_graphs"])):
        d = gdata.Data()
        d.identifiers = torch.from_numpy(gold["%s/%d/identifiers" % (key, g)])
        d.degrees = torch.from_numpy(gold["%s/%d/degrees" % (key, g)])
        out.append(d)
    return out


@pytest.mark.parametrize("enc", ["one_hot_unique", "one_hot_max"])
def test_encode_matches_reference(gold, enc):
    graphs = _source_graphs(gold, "enc_src_" + enc)
    graphs, enc_ids, d_id, enc_deg, d_deg = encoding.encode(graphs, enc, enc, ids={}, degree={})
    assert list(d_id) == gold["enc_%s/d_id" % enc].tolist()
    assert list(d_deg) == gold["enc_%s/d_degree" % enc].tolist()
    for g, d in enumerate(graphs):
        assert np.array_equal(d.identifiers.numpy(), gold["enc_%s/%d/identifiers" % (enc, g)])
        want = gold["enc_%s/%d/degrees" % (enc, g)]
        assert np.array_equal(np.asarray(d.degrees.numpy()).reshape(want.shape), want)
        assert str(d.degrees.dtype) == str(gold["enc_%s/%d/degrees.dtype" % (enc, g)])


def test_encode_without_encoders_keeps_data(gold):
    graphs = _source_graphs(gold, "enc_src_one_hot_unique")
    before = [g.identifiers.clone() for g in graphs]
    graphs, e1, d_id, e2, d_deg = encoding.encode(graphs, None, None)
    assert e1 is None and e2 is None and d_deg == [] and d_id == [1] * before[0].shape[1]
    assert all(torch.equal(a.identifiers, b) for a, b in zip(graphs, before))


@pytest.mark.parametrize("shape", [(1, 1), (5, 3), (4097, 7), (70000, 2), (3000001, 4)])
def test_unique_codes_random(shape):
    rng = np.random.default_rng(shape[0])
    vals = rng.integers(-5, [3 + 40 * c * c for c in range(shape[1])], size=shape)
    vals[:, 0] = rng.integers(0, 3000, size=shape[0]) * 7 - 1000      # sparse range: forces multi-tile scans
    codes, d = encoding.unique_codes(torch.from_numpy(vals))
    for c in range(shape[1]):
        u, inv = np.unique(vals[:, c], return_inverse=True)
        assert d[c] == len(u)
        assert np.array_equal(codes[:, c].numpy(), inv)


def test_unique_codes_edge_cases():
    codes, d = encoding.unique_codes(torch.zeros((0, 3), dtype=torch.int64))
    assert codes.shape == (0, 3) and d == [0, 0, 0]
    codes, d = encoding.unique_codes(torch.tensor([[2.0], [0.0], [2.0]]))
    assert codes[:, 0].tolist() == [1, 0, 1] and d == [2]
    with pytest.raises(NotImplementedError):
        encoding.unique_codes(torch.tensor([[0.5]]))
    with pytest.raises(NotImplementedError):
        encoding.unique_codes(torch.tensor([[0], [1 << 40]]))


def test_one_hot_encoder_matches_reference(gold):
    codes = torch.from_numpy(gold["emb/codes"]).cuda()
    m = encoding.DiscreteEmbedding("one_hot_encoder", 4, gold["emb/d_in"].tolist(), 16)
    assert m.d_out == int(gold["emb/one_hot/d_out"])
    assert np.array_equal(m(codes).cpu().numpy(), gold["emb/one_hot/out"])
    assert len(list(m.parameters())) == 0


@pytest.mark.parametrize("aggr", ["sum", "concat"])
def test_embedding_matches_reference_fwd_bwd(gold, aggr):
    codes = torch.from_numpy(gold["emb/codes"]).cuda()
    m = encoding.DiscreteEmbedding("embedding", 4, gold["emb/d_in"].tolist(), 16, aggr=aggr, init=None)
    sd = {k[len("emb/%s/sd/" % aggr):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("emb/%s/sd/" % aggr)}
    assert set(sd) == set(m.state_dict())
    m.load_state_dict(sd)
    m = m.cuda()
    assert m.d_out == int(gold["emb/%s/d_out" % aggr])
    y = m(codes)
    want = gold["emb/%s/out" % aggr]
    assert np.allclose(y.detach().cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    (y * torch.from_numpy(gold["emb/%s/gy" % aggr]).cuda()).sum().backward()
    for k, p in m.named_parameters():
        assert np.allclose(p.grad.cpu().numpy(), gold["emb/%s/grad/%s" % (aggr, k)], rtol=1e-5, atol=1e-6), k


def test_embedding_out_of_range_raises():
    """A code outside its table: the row comes out NaN at once and the IndexError of the reference's nn.Embedding arrives without a
    host synchronisation per call -- at the next embedding call whose predecessor has completed, or from check_embedding_status()."""
    m = encoding.multi_embedding([3, 3], 4, "sum").cuda()
    for concat in ("sum", "concat"):
        mm = encoding.multi_embedding([3, 3], 4, concat).cuda()
        y = mm(torch.tensor([[0, 1], [0, 3]]).cuda())
        assert torch.isfinite(y[0]).all() and torch.isnan(y[1]).any()
        with pytest.raises(IndexError):
            encoding.check_embedding_status(wait=True)
    y = m(torch.tensor([[0, 3]]).cuda())
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        m(torch.tensor([[0, 1]]).cuda())                 # (the launch before this one has completed: its status is read here)
    encoding.check_embedding_status(wait=True)           # (nothing pending is left behind for other tests)
    encoding.EMBED_STATUS_SYNC = True
    try:
        with pytest.raises(IndexError):
            m(torch.tensor([[0, 3]]).cuda())
    finally:
        encoding.EMBED_STATUS_SYNC = False


def test_ogb_style_encoders_and_linear():
    torch.manual_seed(0)
    a = encoding.DiscreteEmbedding("atom_encoder", 9, None, 32).cuda()
    assert sorted(a.state_dict())[0] == "encoder.atom_embedding_list.0.weight" and len(a.state_dict()) == 9
    x = torch.stack([torch.randint(0, d, (50,)) for d in encoding.ATOM_FEATURE_DIMS], 1).cuda()
    want = sum(a.encoder.atom_embedding_list[i].weight[x[:, i]] for i in range(9))
    assert torch.allclose(a(x), want, rtol=1e-6, atol=1e-6)
    b = encoding.DiscreteEmbedding("bond_encoder", 3, None, 8).cuda()
    assert "encoder.bond_embedding_list.2.weight" in b.state_dict()
    assert b(torch.zeros(5, 3, dtype=torch.long).cuda()).shape == (5, 8)
    oh = encoding.DiscreteEmbedding("atom_one_hot_encoder", 9, None, 0, features_scope="simple")
    assert oh.d_out == 123 and oh(x[:, :2]).sum().item() == 100.0
    lin = encoding.DiscreteEmbedding("linear", 6, None, 10).cuda()
    v = torch.randn(33, 6).cuda().requires_grad_(True)
    y = lin(v)
    ref = torch.nn.functional.linear(v, lin.encoder.weight, lin.encoder.bias)
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5)
    y.sum().backward()
    assert v.grad is not None and lin.encoder.weight.grad is not None
    none = encoding.DiscreteEmbedding("None", 5, None, 0)
    assert none.d_out == 5 and none(torch.arange(4)).shape == (4, 1)
    z = encoding.DiscreteEmbedding("zero_encoder", 1, None, 7)
    assert z(torch.zeros(3, 1).cuda()).shape == (3, 7)
    with pytest.raises(NotImplementedError):
        encoding.DiscreteEmbedding("nope", 1, None, 1)


@pytest.mark.parametrize("rows,dims,d_out,concat", [(300000, [28, 5, 3], 128, False), (150001, [119, 4, 12, 12, 10, 6, 6, 2, 2], 300, False),
                                                    (200000, [7, 3], 64, True), (1000003, [4], 32, False)])
def test_embedding_tables_on_many_rows_fwd_bwd(rows, dims, d_out, concat):
    """gsn_embed_fwd_hip / gsn_embed_bwd_hip (sum or concatenation of per-column embedding tables, utils_graph_learning.py:132-167;
    ogb's Atom / BondEncoder) on 0.15-1 M rows against torch.nn.functional.embedding -- the reference-generated fixtures are a few
    hundred rows; the backward accumulates table slices in per-wave LDS copies and merges them across all workgroups."""
    from gsn_amd.encoding import embed_columns
    g = torch.Generator().manual_seed(rows % 1000 + d_out)
    x = torch.stack([torch.randint(0, n, (rows,), generator=g) for n in dims], 1).cuda()
    tables = [torch.randn(n, d_out, generator=g).cuda().requires_grad_(True) for n in dims]
    y = embed_columns(x, tables, concat)
    parts = [torch.nn.functional.embedding(x[:, c], tables[c].detach().double()) for c in range(len(dims))]
    ref = torch.cat(parts, 1) if concat else sum(parts)
    assert y.shape == ref.shape
    assert float((y.detach().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    w = torch.randn(y.shape, generator=g).cuda()
    (y * w).sum().backward()
    for c, t in enumerate(tables):
        wc = (w[:, c * d_out:(c + 1) * d_out] if concat else w).double()
        gref = torch.zeros(dims[c], d_out, dtype=torch.float64, device="cuda").index_add_(0, x[:, c], wc)
        err = float((t.grad.double() - gref).abs().max())
        assert err <= 2e-5 * float(gref.abs().max()), (c, err, float(gref.abs().max()))


@pytest.mark.skipif(os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") == "1", reason="stream capture cannot free memory without the caching allocator (scripts/oob_check.sh)")
def test_embedding_inside_a_captured_graph():
    """An embedding launch inside a HIP-graph capture (gsn_amd.graphs): the deferred status machinery stays out of the capture, the replay
    equals the eager result on refilled codes."""
    from gsn_amd import graphs
    torch.manual_seed(1)
    m = encoding.multi_embedding([5, 7], 8, "sum").cuda()
    codes = torch.stack([torch.randint(0, 5, (300,)), torch.randint(0, 7, (300,))], 1).cuda()
    with torch.no_grad():
        step = graphs.GraphedStep(lambda: m(codes), warmup=2)
        codes.copy_(torch.stack([torch.randint(0, 5, (300,)), torch.randint(0, 7, (300,))], 1).cuda())
        y = step().clone()
        want = m(codes)
    assert torch.equal(y, want)
    encoding.check_embedding_status(wait=True)


@pytest.mark.parametrize("m_rows,dims,d", [(20000, [37], 300), (9000, [5, 6, 2], 300), (12345, [119, 4, 12, 12, 10, 6, 6, 2, 2], 300),
                                           (5000, [300, 7], 64), (4100, [3], 7), (70000, [448], 128)])
def test_summed_embedding_backward_on_the_matrix_pipe(m_rows, dims, d):
    """gsn_embed_bwd_hip for summed embeddings of >= 4096 rows: OneHot(codes)^T g as bf16 plane products (embed_bwd_mfma_kernel) against an
    fp64 index_add -- tables of 2 .. 448 rows, one to nine code columns, widths that are not multiples of the tile, gradients with a wide
    range of magnitudes; and against the LDS-accumulating kernel through the module's backward at a size below the switch."""
    torch.manual_seed(m_rows)
    m = encoding.multi_embedding(list(dims), d, "sum").cuda()
    codes = torch.stack([torch.randint(0, n, (m_rows,)) for n in dims], 1).cuda()
    if len(dims) == 1:
        codes[: m_rows // 2, 0] = 0                     # (one heavy row: half of the batch lands on it)
    y = m(codes)
    g = (torch.randn(m_rows, d, device="cuda") * torch.logspace(-4, 4, d, device="cuda")).contiguous()
    y.backward(g)
    for c, (name, p) in enumerate(m.named_parameters()):
        ref = torch.zeros(p.shape, dtype=torch.float64, device="cuda").index_add_(0, codes[:, c], g.double())
        bound = torch.zeros(p.shape, dtype=torch.float64, device="cuda").index_add_(0, codes[:, c], g.double().abs())
        err = ((p.grad.double() - ref).abs() / bound.clamp_min(1e-300)).max().item()
        assert err <= 3e-7 * max(1.0, (m_rows / dims[c]) ** 0.5), (name, err)


@pytest.mark.parametrize("m_rows,dims,d,concat", [(32, [1], 300, False), (837, [119, 4, 12, 12, 10, 6, 6, 2, 2], 300, False), (1720, [5, 6, 2], 300, False),
                                                  (1720, [3, 9, 2, 2, 5, 7], 64, True), (300, [500, 7], 32, False), (100, [5000], 16, False), (0, [4, 4], 8, False)])
def test_embedding_backward_with_the_tables_as_launch_arguments(m_rows, dims, d, concat, monkeypatch):
    """gsn_embed_bwd_flat_hip (gradient tables inside one zeroed allocation, base + host offsets as launch arguments: no device pointer array)
    against gsn_embed_bwd_hip on the same inputs and against an fp64 index_add: the LDS-accumulating kernel (few rows), the matrix-pipe
    product (>= 256 rows), concatenated tables, and shapes the flat variant refuses (tables beyond LDS below 256 rows, > 4096 table rows),
    where the module falls back to the pointer-array entry point."""
    from gsn_amd import _abi
    torch.manual_seed(m_rows + d)
    aggr = "concat" if concat else "sum"
    m = encoding.multi_embedding(list(dims), d, aggr).cuda()
    codes = torch.stack([torch.randint(0, n, (m_rows,)) for n in dims], 1).cuda()
    g = torch.randn(m_rows, d * (len(dims) if concat else 1), device="cuda")
    rows = np.ascontiguousarray(dims, dtype=np.int64)
    supported = bool(_abi.lib().gsn_embed_bwd_flat_supported(max(m_rows, 1), len(dims), int(concat), _abi.ptr(rows)))
    assert supported == (len(dims) <= 16 and (sum(dims) <= 448 or (not concat and sum(dims) <= 4096 and m_rows >= 256)))
    got = {}
    for flat in (True, False):
        monkeypatch.setattr(encoding, "EMBED_BWD_FLAT", flat)
        m.zero_grad(set_to_none=True)
        m(codes).backward(g)
        got[flat] = [p.grad.clone() for p in m.parameters()]
    for c, p in enumerate(m.parameters()):
        gc = (g[:, c * d:(c + 1) * d] if concat else g).double()
        ref = torch.zeros(p.shape, dtype=torch.float64, device="cuda").index_add_(0, codes[:, c], gc)
        bound = torch.zeros(p.shape, dtype=torch.float64, device="cuda").index_add_(0, codes[:, c], gc.abs()).clamp_min(1e-300)
        for flat in (True, False):
            err = ((got[flat][c].double() - ref).abs() / bound).max().item()
            assert err <= 3e-7 * max(1.0, (max(m_rows, 1) / dims[c]) ** 0.5), (flat, c, err)
    if supported:
        return
    # asked for a shape it does not handle, the flat entry point refuses and touches nothing
    flat_buf = torch.zeros(sum(n * d for n in dims), device="cuda")
    offs = np.ascontiguousarray(np.cumsum([0] + [n * d for n in dims[:-1]]), dtype=np.int64)
    rc = _abi.lib().gsn_embed_bwd_flat_hip(m_rows, len(dims), d, int(concat), codes.data_ptr(), flat_buf.data_ptr(), _abi.ptr(offs), _abi.ptr(rows),
                                           g.data_ptr(), _abi.current_stream())
    assert rc != 0
    torch.cuda.synchronize()
    assert float(flat_buf.abs().max()) == 0.0

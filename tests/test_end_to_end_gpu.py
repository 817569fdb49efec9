"""End-to-end plumbing check of BASELINE config 1 on the HIP path: the SR(25,12,5,6) isomorphism test of the reference's
README (:82-90, :134): with induced cycle counts k <= 6 as GSN-e identifiers a random-weight 2x64 GSN_sparse network
separates all 105 pairs of the 15 strongly regular graphs (0 % failure); the identifier-free MPNN separates none (100 %).
Counting, identifier encoding, both layers and the sum readout run on our kernels; the glue between them is a few lines of
PyTorch mirroring utils_encoding.one_hot_unique, DiscreteEmbedding('one_hot_encoder') and GNNSubstructures.forward."""
import numpy as np
import pytest
import torch

from helpers import count_case

pytestmark = pytest.mark.gpu


def _sr25_batch():
    c = count_case("sr25_cycle3-6_induced_edge")
    npt, ept, ei = c["node_ptr"], c["edge_ptr"], c["edge_index_local"].copy()
    for g in range(len(npt) - 1):
        ei[:, ept[g]:ept[g + 1]] += npt[g]
    return c, torch.from_numpy(npt), torch.from_numpy(ept), torch.from_numpy(ei)


def test_sr25_isomorphism_test_zero_failures():
    from gsn_amd import layers
    from gsn_amd.counting import CountPlan, count_batch
    c, npt, ept, ei = _sr25_batch()
    dev = "cuda"
    plan = CountPlan.get(c["patterns"], "edge", True)
    ids, _ = count_batch(plan, npt, ept, ei, ids_are_global=True)
    assert np.array_equal(ids.cpu().numpy(), c["counts"])
    # utils_encoding.one_hot_unique: dataset-level dense recoding of every identifier column
    cols, n_classes = [], []
    for j in range(ids.shape[1]):
        u, inv = torch.unique(ids[:, j], return_inverse=True)
        cols.append(inv); n_classes.append(int(u.numel()))
    assert n_classes == [1, 3, 29, 34]          # d_id = [1,3,29,34] for induced cycles k <= 6 on SR25
    idf = layers.one_hot_identifiers(torch.stack(cols, 1), n_classes)
    N = int(npt[-1]); G = len(npt) - 1
    batch = torch.repeat_interleave(torch.arange(G), npt[1:] - npt[:-1]).to(dev)
    ei = ei.to(dev)
    deg = torch.zeros(N, device=dev)

    def embed(seed, use_ids):
        torch.manual_seed(seed)
        base = dict(d_degree=1, degree_as_tag=False, retain_features=True, seed=seed, activation_name="relu", bn=True,
                    msg_kind="general", flow="source_to_target")
        if use_ids:
            l0 = layers.GSN_sparse(d_in=1, d_id=sum(n_classes), id_scope="local", d_msg=64, d_up=64, d_h=[64], **base)
        else:
            l0 = layers.MPNN_sparse(d_in=1, d_msg=64, d_up=64, d_h=[64], **base)
        l1 = layers.MPNN_sparse(d_in=64, d_msg=64, d_up=64, d_h=[64], **base)
        l0, l1 = l0.to(dev).eval(), l1.to(dev).eval()
        x = torch.ones(N, 1, device=dev)
        with torch.no_grad():
            h1 = torch.relu(l0(x, ei, identifiers=idf, degrees=deg))
            h2 = torch.relu(l1(h1, ei, degrees=deg))
            pooled = [layers.global_add_pool_sparse(t, batch, G) for t in (x, h1, h2)]
        return torch.cat(pooled, 1)

    emb = embed(0, True)
    d = torch.pdist(emb.double())
    assert d.numel() == 105
    assert int((d < 1e-2).sum()) == 0, "GSN-e with cycle counts must tell all SR(25,12,5,6) graphs apart"
    emb0 = embed(0, False)
    d0 = torch.pdist(emb0.double())
    scale = emb0.abs().max().item()
    assert int((d0 < 1e-4 * max(scale, 1.0)).sum()) == 105, "plain message passing cannot separate strongly regular graphs"

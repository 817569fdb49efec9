"""count_batch(encoded_rows=False): the int64 counts and the identifier columns of an exact fp16 row pack leave the counting kernel, NO fp32
one-hot rows (gsn_count_encode_pack16_hip with enc_out = NULL); the third result is a Codes object over the counts, tagged with the pack, that the
layers take as ``identifiers``.  Against the three-output launch (utils_ids.py:27 -> DiscreteEmbedding('one_hot_encoder'),
utils_graph_learning.py:170-187): same counts, same pack columns, same layer rows (bit for bit: the layer kernel reads the same pack)."""
import networkx as nx
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(n_graphs, seed):
    from gsn_amd import synth
    b = synth.zinc_shape_batch(n_graphs, seed=seed)
    dev = torch.device("cuda", 0)
    return b, torch.from_numpy(b.edge_index).to(dev), dev


@pytest.mark.parametrize("n_graphs,clamp", [(1, True), (7, True), (300, True), (301, False)])
def test_pack_only_identifiers_equal_the_three_output_launch(n_graphs, clamp):
    from gsn_amd import layers, packs
    from gsn_amd.counting import CountPlan, count_batch
    b, ei, dev = _batch(n_graphs, 40 + n_graphs)
    E = b.num_edges
    plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", True)
    kw = dict(ids_are_global=True, device=dev, check=True)
    pack_a, pack_b = packs.new_edge_pack(E, dev), packs.new_edge_pack(E, dev)
    pack_b[:, 12:] = 0.5                         # (columns the kernel must leave alone)
    out_a, _, enc_a = count_batch(plan, b.node_ptr, b.edge_ptr, ei, encode=([3, 3, 3, 3], clamp), encoded_pack=(pack_a, 0), **kw)
    out_b, _, cd = count_batch(plan, b.node_ptr, b.edge_ptr, ei, encode=([3, 3, 3, 3], clamp), encoded_pack=(pack_b, 0), encoded_rows=False, **kw)
    assert isinstance(cd, layers.Codes) and cd.clamp == clamp and cd.n_classes == [3, 3, 3, 3]
    assert torch.equal(out_a, out_b) and cd.codes.data_ptr() == out_b.data_ptr()
    assert torch.equal(pack_a[:, :12], pack_b[:, :12])
    assert bool((pack_b[:, 12:] == 0.5).all())
    assert torch.equal(pack_b[:, :12].float(), enc_a)           # the pack columns ARE the fp32 rows the other launch wrote
    assert torch.equal(cd.dense(), enc_a)                       # and what the Codes object densifies to, should anything ask
    assert packs._codes_tag(cd) is not None and packs._codes_tag(cd)[0] is pack_b


def test_layer_takes_the_tagged_codes_and_gives_the_same_rows():
    from gsn_amd import layers, packs
    from gsn_amd.counting import CountPlan, count_batch
    b, ei, dev = _batch(200, 9)
    N, E = b.num_nodes, b.num_edges
    plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", True)
    torch.manual_seed(0)
    layer = layers.GSN_edge_sparse(d_in=28, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=128,
                                   d_up=128, d_h=[128], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target").to(dev).eval()
    xc = layers.Codes(torch.from_numpy(b.atom_type).to(dev), [28])
    efc = layers.Codes(torch.from_numpy(b.bond_type).to(dev), [4])
    degrees = torch.zeros(N, device=dev)
    ys = []
    for rows in (True, False):
        ep = packs.new_edge_pack(E, dev)
        _, _, ids = count_batch(plan, b.node_ptr, b.edge_ptr, ei, ids_are_global=True, device=dev, encode=([3, 3, 3, 3], True),
                                encoded_pack=(ep, 0), encoded_rows=rows)
        packs.pack_edge_codes(efc, ep, 12)
        with torch.no_grad():
            ys.append(layer(xc, ei, identifiers=ids, degrees=degrees, edge_features=efc))
    assert torch.equal(ys[0], ys[1])


def test_pack_only_needs_the_counts_and_a_pack():
    from gsn_amd import packs
    from gsn_amd.counting import CountPlan, count_batch
    b, ei, dev = _batch(3, 1)
    plan = CountPlan.get([list(nx.cycle_graph(3).edges)], "edge", True)
    with pytest.raises(ValueError, match="encoded_rows=False"):
        count_batch(plan, b.node_ptr, b.edge_ptr, ei, ids_are_global=True, device=dev, encode=([3], True), encoded_rows=False)
    with pytest.raises(ValueError, match="encoded_rows=False"):
        count_batch(plan, b.node_ptr, b.edge_ptr, ei, ids_are_global=True, device=dev, encode=([3], True), counts=False,
                    encoded_pack=(packs.new_edge_pack(b.num_edges, dev), 0), encoded_rows=False)

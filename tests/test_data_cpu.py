"""Raw-file readers and dataset containers against outputs of the reference's own loaders (utils_data_prep.py) recorded
in tests/golden/dataset.npz by make_golden.py; the raw files under tests/golden/raw are data fixtures."""
import os

import numpy as np
import pytest
import torch

from gsn_amd import data as gdata
from gsn_amd import dataset as gds

HERE = os.path.dirname(os.path.abspath(__file__))
RAW = os.path.join(HERE, "golden", "raw")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "dataset.npz"), allow_pickle=False)


@pytest.mark.parametrize("tag", [False, True])
def test_tu_loader_matches_reference(gold, tag):
    graphs, ncls = gdata.load_data(RAW, "TUTRIM", tag)
    key = "tu_tag%d" % int(tag)
    assert ncls == int(gold[key + "/num_classes"])
    assert len(graphs) == 13
    for g, s in enumerate(graphs):
        assert np.array_equal(s.edge_mat.numpy().reshape(2, -1), gold["%s/%d/edge_mat" % (key, g)]), g
        assert s.edge_mat.dtype == torch.int64
        assert np.array_equal(s.node_features.numpy(), gold["%s/%d/node_features" % (key, g)])
        assert s.label == int(gold["%s/%d/label" % (key, g)])
        assert np.array_equal(np.asarray(s.node_tags), gold["%s/%d/node_tags" % (key, g)])
        assert s.max_neighbor == int(gold["%s/%d/max_neighbor" % (key, g)])


def test_g6_loader_matches_reference(gold):
    graphs, ncls = gdata.load_g6_graphs(RAW, "sr251256")
    assert ncls == int(gold["g6/num_classes"]) == 15
    for g, s in enumerate(graphs):
        assert np.array_equal(s.edge_mat.numpy(), gold["g6/%d/edge_mat" % g])
        assert np.array_equal(s.node_features.numpy(), gold["g6/%d/node_features" % g])
        assert int(s.label) == int(gold["g6/%d/label" % g]) and s.label.dtype == torch.int64


def test_zinc_loader_matches_reference(gold):
    graphs, ncls, nnt, net = gdata.load_zinc_data(os.path.join(RAW, "ZINC"), "ZINC", False)
    assert [len(graphs), ncls, nnt, net] == gold["zinc/meta"].tolist()
    for g, s in enumerate(graphs):
        assert np.array_equal(s.edge_mat.numpy(), gold["zinc/%d/edge_mat" % g])
        assert np.array_equal(s.node_features.numpy(), gold["zinc/%d/node_features" % g])
        assert np.array_equal(s.edge_features.numpy(), gold["zinc/%d/edge_features" % g])
        assert float(s.label) == float(gold["zinc/%d/label" % g])


def test_load_raw_dispatch():
    g, c, a, b = gdata.load_raw(RAW, "sr251256")
    assert len(g) == 15 and c == 15 and a is None and b is None
    g, c, a, b = gdata.load_raw(os.path.join(RAW, "ZINC"), "ZINC")
    assert (c, a, b) == (1, 28, 4)
    g, c, a, b = gdata.load_raw(RAW, "TUTRIM")
    assert len(g) == 13


def test_data_bag_iterates_like_pyg():
    d = gdata.Data()
    d.edge_index = torch.zeros(2, 3, dtype=torch.long)
    d.x = torch.ones(4, 1)
    d.graph_size = 4
    assert d.keys == ["edge_index", "x", "graph_size"]
    assert [k for k, _ in d] == ["edge_index", "x", "graph_size"]
    assert "x" in d and "y" not in d


def test_downgrade_k_slices_identifier_columns(gold):
    n = int(gold["gd_tu_vertex/n_graphs"])
    ds = []
    for g in range(n):
        d = gdata.Data()
        for name in gold["gd_tu_vertex/%d/attr_order" % g].tolist():
            setattr(d, name, torch.from_numpy(np.asarray(gold["gd_tu_vertex/%d/%s" % (g, name)])))
        ds.append(d)
    sizes = gold["gd_tu_vertex/orbit_partition_sizes"].tolist()
    out, osz = gds.downgrade_k(ds, 3, sizes, 3)
    assert osz == gold["downgrade/sizes"].tolist()
    for g, d in enumerate(out):
        assert np.array_equal(d.identifiers.numpy(), gold["downgrade/%d/identifiers" % g])
        assert d.x is ds[g].x and d is not ds[g]


def test_cache_tuple_roundtrip(tmp_path):
    d = gdata.Data(x=torch.ones(3, 1), identifiers=torch.arange(6).reshape(3, 2))
    f = str(tmp_path / "cycle_graph_induced_6.pt")
    gds.save_dataset([d], 2, [1, 1], f)
    graphs, ncls, sizes = gds.load_dataset(f)
    assert ncls == 2 and sizes == [1, 1] and torch.equal(graphs[0].identifiers, d.identifiers)

"""Graph containers and raw-file readers for the preprocessing driver.

Host-side plumbing only: these functions turn the dataset files the reference reads into per-graph tensors with the
same contents *and the same edge order* as the reference's loaders (the row order of GSN-e identifiers follows
``edge_index`` column order, utils_graph_processing.py:141-144, so the order is part of the contract).

* :class:`Data`            -- attribute bag standing in for ``torch_geometric.data.Data`` (utils_data_gen.py:88);
                              iterating yields ``(name, value)`` like PyG's (used by ``downgrade_k``, utils.py:340-342).
* :func:`load_data`        -- TU "powerful-gnns" txt format, utils_data_prep.py:35-136.
* :func:`load_g6_graphs`   -- graph6 files of strongly regular graphs, utils_data_prep.py:197-212.
* :func:`load_zinc_data`   -- benchmarking-gnns ZINC pickles, utils_data_prep.py:139-174.
"""
from __future__ import annotations

import csv
import os
import pickle
from collections import namedtuple

import numpy as np
import torch

from . import synth

SR_DATASETS = ("sr16622", "sr251256", "sr261034", "sr281264", "sr291467", "sr351668", "sr351899", "sr361446", "sr401224")


class Data:
    """Minimal stand-in for PyG's ``Data``: attributes set with ``setattr``; ``keys`` / iteration in insertion order."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def __iter__(self):
        for k in self.keys:
            yield k, getattr(self, k)

    def __contains__(self, key):
        return key in self.__dict__

    def to(self, device):
        for k, v in list(self):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        return self

    def __repr__(self):
        parts = []
        for k, v in self:
            parts.append("%s=%s" % (k, list(v.shape) if isinstance(v, torch.Tensor) else v))
        return "Data(%s)" % ", ".join(parts)


class S2VGraph:
    """Record of the TU loader (fields as utils_data_prep.py:13-32; ``g`` is not kept -- nothing downstream reads it)."""

    def __init__(self, label, node_tags):
        self.label = label
        self.node_tags = node_tags
        self.neighbors = []
        self.node_features = 0
        self.edge_mat = 0
        self.max_neighbor = 0


class _InsertionOrderedGraph:
    """The slice of networkx.Graph semantics the TU loader depends on: nodes and adjacency keep insertion order,
    ``edges()`` lists every undirected edge once, at the endpoint that comes first in node order, and a repeated
    ``add_edge`` does not move anything."""

    def __init__(self):
        self.adj = {}

    def add_node(self, u):
        if u not in self.adj:
            self.adj[u] = {}

    def add_edge(self, u, v):
        self.add_node(u)
        self.add_node(v)
        self.adj[u][v] = True
        self.adj[v][u] = True

    def __len__(self):
        return len(self.adj)

    def edges(self):
        done = set()
        out = []
        for u, nbrs in self.adj.items():
            for v in nbrs:
                if v not in done:
                    out.append((u, v))
            done.add(u)
        return out

    def degrees(self):
        # networkx counts a self loop twice in the degree view
        return [len(nb) + (1 if u in nb else 0) for u, nb in self.adj.items()]


def load_data(path, name, degree_as_tag):
    """TU txt reader -> ``(list of S2VGraph, num_classes)`` (utils_data_prep.py:35-136).

    edge_mat = the graph's undirected edges in insertion order followed by the same list reversed pairwise
    (:104-108); node_features = one-hot of the node tag over the dataset's tag set (:114-125)."""
    g_list, graphs = [], []
    label_dict, feat_dict = {}, {}
    with open("%s/%s.txt" % (path, name), "r") as f:
        n_g = int(f.readline().strip())
        for _ in range(n_g):
            n, l = [int(w) for w in f.readline().strip().split()]
            if l not in label_dict:
                label_dict[l] = len(label_dict)
            g = _InsertionOrderedGraph()
            node_tags = []
            for j in range(n):
                g.add_node(j)
                row = f.readline().strip().split()
                tmp = int(row[1]) + 2
                row = [int(w) for w in row[:tmp]] if tmp != len(row) else [int(w) for w in row]
                if row[0] not in feat_dict:
                    feat_dict[row[0]] = len(feat_dict)
                node_tags.append(feat_dict[row[0]])
                for k in range(2, len(row)):
                    g.add_edge(j, row[k])
            assert len(g) == n
            g_list.append(S2VGraph(l, node_tags))
            graphs.append(g)
    for s, g in zip(g_list, graphs):
        order = list(g.adj.keys())
        s.neighbors = [[] for _ in range(len(g))]
        edges = g.edges()
        for i, j in edges:
            s.neighbors[i].append(j)
            s.neighbors[j].append(i)
        s.max_neighbor = max(len(nb) for nb in s.neighbors) if len(g) else 0
        s.label = label_dict[s.label]
        pairs = [list(p) for p in edges]
        pairs.extend([[i, j] for j, i in pairs])
        s.edge_mat = torch.LongTensor(pairs).transpose(0, 1) if pairs else torch.zeros((2, 0), dtype=torch.long)
        if degree_as_tag:
            deg = dict(zip(order, g.degrees()))
            s.node_tags = [deg[u] for u in order]
    tagset = set([])
    for s in g_list:
        tagset = tagset.union(set(s.node_tags))
    tagset = list(tagset)
    tag2index = {tagset[i]: i for i in range(len(tagset))}
    for s in g_list:
        s.node_features = torch.zeros(len(s.node_tags), len(tagset))
        s.node_features[range(len(s.node_tags)), [tag2index[t] for t in s.node_tags]] = 1
    return g_list, len(label_dict)


def load_g6_graphs(path, name):
    """graph6 reader -> ``(list of Graph(node_features, edge_mat, label), num_classes)``: x = ones [n,1], edge_mat =
    both directions, coalesced (sorted by row then column), label = index in the file (utils_data_prep.py:197-212)."""
    Graph = namedtuple("Graph", ["node_features", "edge_mat", "label"])
    out = []
    for i, (n, und) in enumerate(synth.read_graph6_file(os.path.join(path, name + ".g6"))):
        ei = torch.from_numpy(synth.undirected_to_edge_index(n, und, sort=True))
        out.append(Graph(torch.ones(n, 1), ei, torch.tensor(i).long()))
    return out, len(out)


def load_zinc_data(path, name, degree_as_tag, num_atom_type=28, num_bond_type=4):
    """ZINC pickles (lists of dicts with ``atom_type`` [n], dense ``bond_type`` [n,n], ``logP_SA_cycle_normalized``)
    -> ``(graphs, 1, num_atom_type, num_bond_type)``; edge_mat = nonzero(adj) in row-major order, edge_features = the
    bond types at those positions (utils_data_prep.py:139-174)."""
    assert name.upper() == "ZINC"
    Graph = namedtuple("Graph", ["node_features", "edge_mat", "edge_features", "label"])
    data = []
    for split in ("train", "val", "test"):
        with open(os.path.join(path, "molecules", "%s.pickle" % split), "rb") as f:
            mols = pickle.load(f)
        with open(os.path.join(path, "indices", "%s.index" % split), "r") as f:
            idx = [list(map(int, row)) for row in csv.reader(f)]
        for i in idx[0]:
            m = mols[i]
            adj = m["bond_type"]
            el = (adj != 0).nonzero()
            data.append(Graph(m["atom_type"].long(), el.permute(1, 0), adj[el[:, 0], el[:, 1]].reshape(-1).long(),
                              m["logP_SA_cycle_normalized"]))
    return data, 1, num_atom_type, num_bond_type


def load_ogb_data(path, name, degree_as_tag):
    """ogb datasets need the ``ogb`` package to read their raw files (utils_data_prep.py:177-194)."""
    try:
        from ogb.graphproppred import PygGraphPropPredDataset  # noqa: F401
    except ImportError as e:
        raise ImportError("load_ogb_data needs the 'ogb' package, which is not installed here") from e
    transform = None
    if name == "ogbg-ppa":
        def transform(d):
            d.x = torch.zeros(d.num_nodes, dtype=torch.long)
            return d
    dataset = PygGraphPropPredDataset(name=name, root=path, transform=transform)
    Graph = namedtuple("Graph", ["node_features", "edge_mat", "edge_features", "label"])
    out = [Graph(d.x, d.edge_index, d.edge_attr, d.y) for d in dataset]
    return out, (dataset.num_classes if name == "ogbg-ppa" else dataset.num_tasks)


def load_raw(data_path, dataset_name):
    """Loader dispatch of generate_dataset (utils_data_gen.py:46-56) -> (graphs, num_classes, num_node_type, num_edge_type)."""
    if "ogb" in data_path:
        g, c = load_ogb_data(data_path, dataset_name, False)
        return g, c, None, None
    if dataset_name == "ZINC":
        return load_zinc_data(data_path, dataset_name, False)
    if dataset_name in SR_DATASETS:
        g, c = load_g6_graphs(data_path, dataset_name)
        return g, c, None, None
    g, c = load_data(data_path, dataset_name, False)
    return g, c, None, None

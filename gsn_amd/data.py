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


def _tu_records(path, name):
    """Tokenise a TU "powerful-gnns" file once: yields ``(label, tags, neighbour_lists)`` per graph.  A node line is
    ``tag m n_1 .. n_m`` optionally followed by attribute columns the reference drops (utils_data_prep.py:58-66)."""
    with open(os.path.join(path, name + ".txt"), "r") as fh:
        lines = iter(fh.read().split("\n"))
    n_graphs = int(next(lines).split()[0])
    for _ in range(n_graphs):
        n_nodes, label = (int(w) for w in next(lines).split()[:2])
        tags, nbrs = [], []
        for _ in range(n_nodes):
            tok = next(lines).split()
            m = int(tok[1])
            tags.append(int(tok[0]))
            nbrs.append([int(w) for w in tok[2:2 + m]])
        yield label, tags, nbrs


def _undirected_in_first_touch_order(nbrs):
    """What the reference gets out of networkx for one graph (utils_data_prep.py:52-72, :104-108), without networkx: vertices in the order
    they are first touched (vertex j when its line is read, a neighbour when the first line naming it is read), every vertex's
    adjacency in the order its edges first appear, each undirected edge listed once at whichever endpoint comes first in vertex order;
    a self loop counts twice in the degree.  -> (vertex order, edge list, degree per vertex in vertex order)"""
    rank, adjacency = {}, []

    def touch(u):
        if u not in rank:
            rank[u] = len(adjacency)
            adjacency.append({})
        return rank[u]

    for j, row in enumerate(nbrs):
        rj = touch(j)
        for v in row:
            rv = touch(v)
            adjacency[rj].setdefault(v, None)
            adjacency[rv].setdefault(j, None)
    order = sorted(rank, key=rank.get)
    edges = [(u, v) for u in order for v in adjacency[rank[u]] if rank[v] >= rank[u]]
    degrees = [len(adjacency[rank[u]]) + (1 if u in adjacency[rank[u]] else 0) for u in order]
    return order, edges, degrees


def _first_seen(values):
    """value -> index in order of first appearance"""
    table = {}
    for v in values:
        table.setdefault(v, len(table))
    return table


def load_data(path, name, degree_as_tag):
    """TU txt reader -> ``(list of S2VGraph, num_classes)`` (utils_data_prep.py:35-136).

    edge_mat = the graph's undirected edges in first-touch order followed by the same list reversed pairwise
    (:104-108); node_features = one-hot of the node tag over the dataset's tag set (:114-125)."""
    records = list(_tu_records(path, name))
    class_of = _first_seen(label for label, _, _ in records)
    tag_of = _first_seen(t for _, tags, _ in records for t in tags)
    out = []
    for label, tags, nbrs in records:
        order, edges, degrees = _undirected_in_first_touch_order(nbrs)
        if len(order) != len(nbrs):
            raise AssertionError("%s: a neighbour id outside 0 .. %d" % (name, len(nbrs) - 1))
        g = S2VGraph(class_of[label], degrees if degree_as_tag else [tag_of[t] for t in tags])
        g.neighbors = [[] for _ in order]
        for u, v in edges:
            g.neighbors[u].append(v)
            g.neighbors[v].append(u)
        g.max_neighbor = max((len(nb) for nb in g.neighbors), default=0)
        both = edges + [(v, u) for u, v in edges]
        g.edge_mat = torch.tensor(both, dtype=torch.long).t().contiguous() if both else torch.zeros((2, 0), dtype=torch.long)
        out.append(g)
    # the feature COLUMN of a tag is its position in CPython's iteration order of the set grown by these unions (:114-118) -- not sorted
    # order once the tags (degrees) exceed the table size, so the set is grown the same way
    seen = set()
    for g in out:
        seen = seen.union(set(g.node_tags))
    column = {t: i for i, t in enumerate(seen)}
    for g in out:
        idx = torch.tensor([column[t] for t in g.node_tags], dtype=torch.long)
        g.node_features = torch.zeros(len(g.node_tags), len(column))
        g.node_features[torch.arange(len(g.node_tags)), idx] = 1
    return out, len(class_of)


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

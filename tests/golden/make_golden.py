#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python.

Runs only in the build container (needs /root/reference); the GPU box and the tests read the
committed ``*.npz`` files, never the reference.  Nothing from the reference is copied: its modules
are imported from where they lie, with two things it cannot import here replaced by stand-ins
that live in this script:

* ``graph_tool`` (C++/Boost, not installable here) -> a small networkx-backed module exposing only
  the surface utils_graph_processing.py touches.  Enumeration is done by networkx 3.4.2's VF2
  (``GraphMatcher.subgraph_monomorphisms_iter`` for induced=False, ``subgraph_isomorphisms_iter``
  for induced=True), i.e. an implementation independent of both graph-tool and our oracle.
* ``torch_geometric.utils`` / ``ogb`` -> the handful of helpers the hot-path files import
  (semantics per PyG >= 1.4.3 docs: remove_self_loops keeps order, to_undirected = concat both
  directions + coalesce (sort by row*N+col, unique), degree = scatter-add of ones).

Because graph-tool itself is absent, parity is pinned to "reference Python semantics over an
independent VF2", cross-checked against closed forms for strongly regular graphs (see
tests/test_oracle_golden.py).  Usage:  python tests/golden/make_golden.py [--only orbits,counts,layers]
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import types
import warnings

import numpy as np
import torch
import networkx as nx
from networkx.algorithms import isomorphism as nxiso

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from gsn_amd import synth  # noqa: E402  (ours)


# ----------------------------------------------------------------------------------------------
# stand-in for graph_tool (only what the reference's hot path touches)
# ----------------------------------------------------------------------------------------------
class _Map:
    """What graph-tool yields per match: a vertex property map indexed by pattern vertex."""

    def __init__(self, arr):
        self._a = np.asarray(arr, dtype=np.int32)

    def __iter__(self):
        return iter(self._a.tolist())

    def __len__(self):
        return len(self._a)

    def get_array(self):
        return self._a


class _Graph:
    def __init__(self, directed=True):
        self.directed = directed
        self._edges = []
        self._n = 0

    def add_edge_list(self, edge_list):
        for e in edge_list:
            u, v = int(e[0]), int(e[1])
            self._edges.append((u, v))
            self._n = max(self._n, u + 1, v + 1)

    def get_edges(self):
        return np.asarray(self._edges, dtype=np.int64).reshape(-1, 2)

    def get_vertices(self):
        return np.arange(self._n)

    def to_nx(self):
        g = nx.DiGraph() if self.directed else nx.Graph()
        g.add_nodes_from(range(self._n))
        g.add_edges_from(self._edges)
        return g


def _remove_self_loops(g):
    g._edges = [(u, v) for (u, v) in g._edges if u != v]


def _remove_parallel_edges(g):
    seen = set()
    out = []
    for (u, v) in g._edges:
        key = (u, v) if g.directed else (min(u, v), max(u, v))
        if key not in seen:
            seen.add(key)
            out.append((u, v))
    g._edges = out


def _subgraph_isomorphism(sub, g, max_n=0, vertex_label=None, edge_label=None, induced=False,
                          subgraph=True, generator=False):
    assert subgraph
    H, G = sub.to_nx(), g.to_nx()
    k = H.number_of_nodes()
    gm = (nxiso.DiGraphMatcher if g.directed else nxiso.GraphMatcher)(G, H)
    it = gm.subgraph_isomorphisms_iter() if induced else gm.subgraph_monomorphisms_iter()

    def gen():
        for m in it:  # dict: G node -> H node
            arr = np.empty(k, dtype=np.int32)
            for gv, hv in m.items():
                arr[hv] = gv
            yield _Map(arr)

    return gen() if generator else list(gen())


def install_stubs():
    gt = types.ModuleType("graph_tool")
    gt.Graph = _Graph
    gt.stats = types.ModuleType("graph_tool.stats")
    gt.stats.remove_self_loops = _remove_self_loops
    gt.stats.remove_parallel_edges = _remove_parallel_edges
    gt.topology = types.ModuleType("graph_tool.topology")
    gt.topology.subgraph_isomorphism = _subgraph_isomorphism
    sys.modules["graph_tool"] = gt
    sys.modules["graph_tool.stats"] = gt.stats
    sys.modules["graph_tool.topology"] = gt.topology

    tg = types.ModuleType("torch_geometric")
    tgu = types.ModuleType("torch_geometric.utils")

    def remove_self_loops(edge_index, edge_attr=None):
        mask = edge_index[0] != edge_index[1]
        return edge_index[:, mask], (None if edge_attr is None else edge_attr[mask])

    def to_undirected(edge_index, num_nodes=None):
        n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
        row, col = edge_index
        row, col = torch.cat([row, col]), torch.cat([col, row])
        key = torch.unique(row * n + col)  # sorted
        return torch.stack([key // n, key % n], 0)

    def degree(index, num_nodes=None, dtype=None):
        n = int(index.max()) + 1 if num_nodes is None else num_nodes
        out = torch.zeros(n, dtype=dtype or torch.float)
        return out.scatter_add_(0, index, torch.ones_like(index, dtype=out.dtype))

    def is_undirected(edge_index, *a, **k):
        return True

    tgu.remove_self_loops = remove_self_loops
    tgu.to_undirected = to_undirected
    tgu.degree = degree
    tgu.is_undirected = is_undirected
    tgu.sort_edge_index = lambda ei, *a, **k: ei          # imported by utils_data_gen.py:3, never called
    tg.utils = tgu
    tgd = types.ModuleType("torch_geometric.data")

    class Data:                                           # attribute bag; iteration as PyG 1.x (sorted keys)
        def __init__(self, **kw):
            for k_, v_ in kw.items():
                setattr(self, k_, v_)

        @property
        def keys(self):
            return [k_ for k_ in self.__dict__ if self.__dict__[k_] is not None]

        def __iter__(self):
            for k_ in sorted(self.keys):
                yield k_, getattr(self, k_)

    tgd.Data = Data
    tg.data = tgd
    sys.modules["torch_geometric"] = tg
    sys.modules["torch_geometric.utils"] = tgu
    sys.modules["torch_geometric.data"] = tgd

    ogb = types.ModuleType("ogb")
    gpp = types.ModuleType("ogb.graphproppred")
    me = types.ModuleType("ogb.graphproppred.mol_encoder")
    me.AtomEncoder = me.BondEncoder = object
    gpp.PygGraphPropPredDataset = object                  # imported by utils_data_prep.py:10, not used here
    ou = types.ModuleType("ogb.utils")
    of = types.ModuleType("ogb.utils.features")
    of.get_atom_feature_dims = lambda: [119, 4, 12, 12, 10, 6, 6, 2, 2]
    of.get_bond_feature_dims = lambda: [5, 6, 2]
    for name, m in [("ogb", ogb), ("ogb.graphproppred", gpp), ("ogb.graphproppred.mol_encoder", me),
                    ("ogb.utils", ou), ("ogb.utils.features", of)]:
        sys.modules[name] = m


def import_reference():
    install_stubs()
    sys.path.insert(1, REF)
    import importlib
    mods = {}
    for name in ["utils_graph_processing", "utils_ids", "models_misc", "utils_graph_learning", "utils_data_prep",
                 "utils_data_gen", "utils_encoding", "utils"]:
        mods[name] = importlib.import_module(name)
    for name in ["GSN_sparse", "GSN_edge_sparse", "MPNN_sparse", "MPNN_edge_sparse",
                 "GSN_edge_sparse_ogb", "MPNN_edge_sparse_ogb"]:
        mods[name] = importlib.import_module("graph_filters." + name)
    for m in mods.values():
        assert m.__file__.startswith(REF), m.__file__
    return mods


# ----------------------------------------------------------------------------------------------
# pattern families (how utils.get_custom_edge_list builds them: networkx generators / read_graph6,
# utils.py:16-33) -- edge lists "as networkx orders them"
# ----------------------------------------------------------------------------------------------
def pattern_families():
    fams = {}
    fams["cycle_graph"] = [list(nx.cycle_graph(k).edges) for k in range(3, 9)]
    fams["complete_graph"] = [list(nx.complete_graph(k).edges) for k in range(3, 7)]
    fams["path_graph"] = [list(nx.path_graph(k).edges) for k in range(3, 7)]
    fams["star_graph"] = [list(nx.star_graph(k).edges) for k in range(2, 6)]
    fams["binomial_tree"] = [list(nx.binomial_tree(k).edges) for k in range(2, 4)]
    fams["diamond_graph"] = [list(nx.diamond_graph().edges)]
    fams["nonisomorphic_trees"] = [list(g.edges) for k in range(3, 7) for g in nx.nonisomorphic_trees(k)]
    for k in range(3, 7):
        gs = nx.read_graph6(os.path.join(REF, "datasets/all_simple_graphs/graph%dc.g6" % k))
        gs = gs if isinstance(gs, list) else [gs]
        fams["all_simple_graphs_%d" % k] = [list(g.edges) for g in gs]
    return fams


def pack_edge_lists(edge_lists):
    """ragged list of edge lists -> (ptr, flat [m,2])"""
    ptr = np.cumsum([0] + [len(e) for e in edge_lists]).astype(np.int64)
    flat = np.asarray([p for e in edge_lists for p in e], dtype=np.int64).reshape(-1, 2)
    return ptr, flat


def gen_orbits(ref, out):
    ugp = ref["utils_graph_processing"]
    fams = pattern_families()
    rec = {}
    names = []
    import io
    import contextlib
    for fam, edge_lists in fams.items():
        for pi, el in enumerate(edge_lists):
            key = "%s/%d" % (fam, pi)
            names.append(key)
            with contextlib.redirect_stdout(io.StringIO()):
                _, part, memb, aut = ugp.automorphism_orbits(edge_list=el, print_msgs=False, directed=False,
                                                             directed_orbits=False)
                k = len(memb)
                rec[key + "/edges"] = np.asarray(el, dtype=np.int64).reshape(-1, 2)
                rec[key + "/v_membership"] = np.asarray([memb[v] for v in range(k)], dtype=np.int64)
                rec[key + "/aut_count"] = np.int64(aut)
                rec[key + "/n_vorbits"] = np.int64(len(part))
                for dirorb in (False, True):
                    _, epart, ememb, aut2 = ugp.induced_edge_automorphism_orbits(edge_list=el, directed=False,
                                                                                 directed_orbits=dirorb)
                    assert aut2 == aut
                    sfx = "_dir" if dirorb else ""
                    rec[key + "/e_membership" + sfx] = np.asarray([ememb[i] for i in range(len(ememb))], dtype=np.int64)
                    rec[key + "/n_eorbits" + sfx] = np.int64(len(epart))
                    # the sorted bidirectional edge list the membership is indexed by: recover it from the partition
                    elist = [None] * len(ememb)
                    seen = {}
                    for orb, edges in epart.items():
                        for e in edges:
                            seen.setdefault(orb, []).append(tuple(int(x) for x in e))
                    # positions: edges were visited in sorted order; rebuild by sorting all directed edges
                    alle = sorted(e for edges in seen.values() for e in edges)
                    rec[key + "/e_list" + sfx] = np.asarray(alle, dtype=np.int64).reshape(-1, 2)
                # deprecated line-graph variant (utils_graph_processing.py:189) for a few small patterns
                if fam in ("cycle_graph", "complete_graph", "path_graph", "star_graph", "binomial_tree", "diamond_graph",
                           "nonisomorphic_trees", "all_simple_graphs_3", "all_simple_graphs_4", "all_simple_graphs_5") and len(el) > 1:
                    _, lpart, lmemb, aut3 = ugp.edge_automorphism_orbits(edge_list=el, directed=False)
                    assert aut3 == aut
                    rec[key + "/line_membership"] = np.asarray([lmemb[i] for i in range(len(lmemb))], dtype=np.int64)
                    rec[key + "/line_n_orbits"] = np.int64(len(lpart))
                    # the partition as rows (orbit, u, v): the line graph's node tuples in the order the reference lists them
                    rec[key + "/line_partition"] = np.asarray([(o, e[0], e[1]) for o in sorted(lpart) for e in lpart[o]],
                                                              dtype=np.int64).reshape(-1, 3)
    rec["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(out, "orbits.npz"), **rec)
    print("orbits: %d patterns" % len(names))


# ----------------------------------------------------------------------------------------------
# counting cases
# ----------------------------------------------------------------------------------------------
def sr25_graphs():
    gs = nx.read_graph6(os.path.join(REF, "datasets/SR_graphs/sr251256/sr251256.g6"))
    out = []
    for g in gs:
        n = g.number_of_nodes()
        out.append((n, synth.undirected_to_edge_index(n, list(g.edges()))))  # = to_undirected(edges), sorted
    return out


def imdb_graphs():
    """edge_mat exactly as the reference's TU loader builds it (utils_data_prep.py:58-110): networkx
    graph filled row by row, then g.edges() followed by the reversed pairs."""
    out = []
    with open(os.path.join(REF, "datasets/social/IMDBBINARY/IMDBBINARY.txt")) as f:
        n_g = int(f.readline())
        for _ in range(n_g):
            n, _l = (int(w) for w in f.readline().split())
            g = nx.Graph()
            for j in range(n):
                g.add_node(j)
                row = [int(w) for w in f.readline().split()]
                for k in row[2:2 + row[1]]:
                    g.add_edge(j, k)
            edges = [list(p) for p in g.edges()]
            edges.extend([[i, j] for j, i in edges])
            ei = np.asarray(edges, dtype=np.int64).reshape(-1, 2).T
            out.append((n, np.ascontiguousarray(ei)))
    return out


def run_counts(ref, graphs, edge_lists, mode, induced, directed_orbits=False, num_nodes=None):
    """-> list of int64 arrays [rows, sum orbits] via the reference's subgraph_counts2ids-style loop
    (count_fn per pattern, concat, .long()), calling the reference functions directly."""
    import io
    import contextlib
    ugp = ref["utils_graph_processing"]
    dicts = []
    with contextlib.redirect_stdout(io.StringIO()):
        for el in edge_lists:
            fn = ugp.automorphism_orbits if mode == "vertex" else ugp.induced_edge_automorphism_orbits
            sg, part, memb, aut = fn(edge_list=el, directed=False, directed_orbits=directed_orbits)
            dicts.append({"subgraph": sg, "orbit_partition": part, "orbit_membership": memb, "aut_count": aut})
    cfn = ugp.subgraph_isomorphism_vertex_counts if mode == "vertex" else ugp.subgraph_isomorphism_edge_counts
    outs = []
    for gi, (n, ei) in enumerate(graphs):
        nn = n if num_nodes is None else num_nodes[gi]
        ids = None
        for d in dicts:
            c = cfn(torch.from_numpy(ei), subgraph_dict=d, induced=induced, num_nodes=nn, directed=False)
            assert c.dtype == torch.float64
            ids = c if ids is None else torch.cat((ids, c), 1)
        outs.append(ids.long().numpy())
    return outs


def save_case(rec, name, graphs, edge_lists, mode, induced, outs, directed_orbits=False, num_nodes=None):
    b = synth.collate([(n if num_nodes is None else num_nodes[i], ei) for i, (n, ei) in enumerate(graphs)])
    pptr, pflat = pack_edge_lists(edge_lists)
    rec[name + "/node_ptr"] = b.node_ptr
    rec[name + "/edge_ptr"] = b.edge_ptr
    rec[name + "/edge_index_local"] = np.concatenate([ei for _, ei in graphs], axis=1) if graphs else np.zeros((2, 0), np.int64)
    rec[name + "/pattern_ptr"] = pptr
    rec[name + "/pattern_edges"] = pflat
    rec[name + "/mode"] = np.asarray(mode)
    rec[name + "/induced"] = np.bool_(induced)
    rec[name + "/directed_orbits"] = np.bool_(directed_orbits)
    rec[name + "/counts"] = np.concatenate(outs, axis=0) if outs else np.zeros((0, 0), np.int64)
    rec.setdefault("names", []).append(name)


def gen_counts(ref, out):
    rec = {}
    t0 = time.time()
    cyc = lambda ks: [list(nx.cycle_graph(k).edges) for k in ks]
    clq = lambda ks: [list(nx.complete_graph(k).edges) for k in ks]

    # --- SR(25,12,5,6): BASELINE config 1.  induced (README.md:84) for all 15 graphs; non-induced for 2.
    sr = sr25_graphs()
    for mode in ("vertex", "edge"):
        o = run_counts(ref, sr, cyc(range(3, 7)), mode, True)
        save_case(rec, "sr25_cycle3-6_induced_%s" % mode, sr, cyc(range(3, 7)), mode, True, o)
        print("sr25 induced", mode, "%.0fs" % (time.time() - t0), flush=True)
    for mode in ("vertex", "edge"):
        o = run_counts(ref, sr[:2], cyc(range(3, 6)), mode, False)
        save_case(rec, "sr25_cycle3-5_mono_%s" % mode, sr[:2], cyc(range(3, 6)), mode, False, o)
        print("sr25 mono", mode, "%.0fs" % (time.time() - t0), flush=True)
    o = run_counts(ref, sr[:1], cyc([6]), "edge", False)
    save_case(rec, "sr25_cycle6_mono_edge_g0", sr[:1], cyc([6]), "edge", False, o)
    print("sr25 C6 mono", "%.0fs" % (time.time() - t0), flush=True)

    # --- IMDB-BINARY (config 3): cliques k<=5, GSN-v (vertex) and GSN-e (edge), non-induced (README.md:99)
    imdb = imdb_graphs()
    sizes = np.array([n for n, _ in imdb])
    sel = [i for i in range(24) if i not in (8, 11, 15)] + [int(i) for i in np.argsort(-sizes)[2:5]]  # skip the 3 with >1M K5 maps (networkx speed)
    sub = [imdb[i] for i in sel]
    for mode in ("vertex", "edge"):
        o = run_counts(ref, sub, clq(range(3, 6)), mode, False)
        save_case(rec, "imdb_clique3-5_mono_%s" % mode, sub, clq(range(3, 6)), mode, False, o)
        print("imdb", mode, "%.0fs" % (time.time() - t0), flush=True)
    big = [imdb[int(np.argmax(sizes))]]
    o = run_counts(ref, big, clq([3, 4]), "vertex", False)
    save_case(rec, "imdb_largest_clique3-4_mono_vertex", big, clq([3, 4]), "vertex", False, o)
    print("imdb largest n=%d E=%d" % (big[0][0], big[0][1].shape[1]), "%.0fs" % (time.time() - t0), flush=True)

    # --- ZINC-shape synthetic molecules (config 2): cycles 3..6 (and up to 8 for GSN-v as README.md:112)
    zb = synth.zinc_shape_batch(48, seed=0)
    zg = [zb.graph(g) for g in range(zb.num_graphs)]
    for mode, ks in (("edge", range(3, 7)), ("vertex", range(3, 9))):
        for induced in (False, True):
            o = run_counts(ref, zg, cyc(ks), mode, induced)
            save_case(rec, "zinc48_cycle%d-%d_%s_%s" % (ks[0], ks[-1], "induced" if induced else "mono", mode),
                      zg, cyc(ks), mode, induced, o)
    print("zinc", "%.0fs" % (time.time() - t0), flush=True)

    # --- ER graphs with the 21 connected 5-vertex patterns (config 5) + k=3,4 families
    g5 = nx.read_graph6(os.path.join(REF, "datasets/all_simple_graphs/graph5c.g6"))
    pats5 = [list(g.edges) for g in g5]
    pats345 = []
    for k in (3, 4, 5):
        gs = nx.read_graph6(os.path.join(REF, "datasets/all_simple_graphs/graph%dc.g6" % k))
        pats345 += [list(g.edges) for g in (gs if isinstance(gs, list) else [gs])]
    er = [synth.er_graph(8, 12, 1), synth.er_graph(16, 40, 2), synth.er_graph(32, 80, 3), synth.er_graph(20, 60, 4)]
    for mode in ("vertex", "edge"):
        for induced in (False, True):
            o = run_counts(ref, er, pats345, mode, induced)
            save_case(rec, "er_small_allsimple3-5_%s_%s" % ("induced" if induced else "mono", mode),
                      er, pats345, mode, induced, o)
        print("er small", mode, "%.0fs" % (time.time() - t0), flush=True)
    er128 = [synth.er_graph(128, 260, 5), synth.er_graph(70, 200, 6)]
    for mode in ("vertex", "edge"):
        o = run_counts(ref, er128, pats5, mode, True)
        save_case(rec, "er128_allsimple5_induced_%s" % mode, er128, pats5, mode, True, o)
        print("er128", mode, "%.0fs" % (time.time() - t0), flush=True)
    # directed_orbits=True variant (utils_graph_processing.py:78-79)
    o = run_counts(ref, er[:3], pats345, "edge", False, directed_orbits=True)
    save_case(rec, "er_small_allsimple3-5_mono_edge_dirorb", er[:3], pats345, "edge", False, o, directed_orbits=True)

    # --- other families: paths, stars, trees (non-induced and induced) on small graphs
    fam = [list(nx.path_graph(k).edges) for k in (3, 4, 5, 6)] + [list(nx.star_graph(k).edges) for k in (2, 3, 4)] \
        + [list(nx.diamond_graph().edges)] + [list(nx.binomial_tree(3).edges)]
    small = [synth.er_graph(12, 20, 7), synth.er_graph(14, 30, 8)] + zg[:4]
    for mode in ("vertex", "edge"):
        for induced in (False, True):
            o = run_counts(ref, small, fam, mode, induced)
            save_case(rec, "mixed_families_%s_%s" % ("induced" if induced else "mono", mode), small, fam, mode, induced, o)
    print("families", "%.0fs" % (time.time() - t0), flush=True)

    # --- edge cases (fidelity checklist): trailing isolated vertices, duplicate columns,
    #     self loops passed straight to the count fn, unsorted columns, an edgeless graph (vertex mode)
    rng = np.random.default_rng(11)
    n, ei = synth.er_graph(10, 18, 9)
    perm = rng.permutation(ei.shape[1])
    ei_shuf = ei[:, perm]
    dup = np.concatenate([ei_shuf, ei_shuf[:, :7], ei_shuf[:, 3:5]], axis=1)           # duplicates: last wins
    loops = np.concatenate([ei_shuf[:, :5], np.array([[2, 7], [2, 7]]), ei_shuf[:, 5:]], axis=1)  # self loops inside
    graphs = [(n, ei_shuf), (n, dup), (n, loops), (n, ei)]
    nn = [n, n, n, n + 3]
    pats = cyc([3, 4]) + [list(nx.path_graph(3).edges)]
    for mode in ("vertex", "edge"):
        o = run_counts(ref, graphs, pats, mode, False, num_nodes=nn)
        save_case(rec, "edge_cases_%s" % mode, graphs, pats, mode, False, o, num_nodes=nn)
    empty = [(5, np.zeros((2, 0), dtype=np.int64))]
    o = run_counts(ref, empty, pats, "vertex", False)
    save_case(rec, "empty_graph_vertex", empty, pats, "vertex", False, o)

    rec["names"] = np.asarray(rec["names"])
    np.savez_compressed(os.path.join(out, "counts.npz"), **rec)
    print("counts: %d cases, %.0fs" % (len(rec["names"]), time.time() - t0))

    # --- subgraph_counts2ids end to end (utils_ids.py:7-29) incl. self-loop stripping with edge_features
    uid = ref["utils_ids"]
    ugp = ref["utils_graph_processing"]
    rec2 = {}
    import io
    import contextlib
    for mode in ("vertex", "edge"):
        with contextlib.redirect_stdout(io.StringIO()):
            dicts = []
            for el in cyc([3, 4, 5]):
                fn = ugp.automorphism_orbits if mode == "vertex" else ugp.induced_edge_automorphism_orbits
                sg, part, memb, aut = fn(edge_list=el, directed=False, directed_orbits=False)
                dicts.append({"subgraph": sg, "orbit_partition": part, "orbit_membership": memb, "aut_count": aut})
        data = types.SimpleNamespace()
        data.x = torch.ones(n + 2, 1)
        data.edge_index = torch.from_numpy(loops.copy())
        data.edge_features = torch.arange(loops.shape[1]) + 100
        cfn = ugp.subgraph_isomorphism_vertex_counts if mode == "vertex" else ugp.subgraph_isomorphism_edge_counts
        res = uid.subgraph_counts2ids(cfn, data, dicts, {"induced": False, "directed": False})
        rec2[mode + "/in_edge_index"] = loops
        rec2[mode + "/in_num_nodes"] = np.int64(n + 2)
        rec2[mode + "/out_edge_index"] = res.edge_index.numpy()
        rec2[mode + "/out_edge_features"] = res.edge_features.numpy()
        rec2[mode + "/identifiers"] = res.identifiers.numpy()
        assert res.identifiers.dtype == torch.int64
    pptr, pflat = pack_edge_lists(cyc([3, 4, 5]))
    rec2["pattern_ptr"], rec2["pattern_edges"] = pptr, pflat
    np.savez_compressed(os.path.join(out, "counts2ids.npz"), **rec2)


# ----------------------------------------------------------------------------------------------
# HP-2 layer vectors
# ----------------------------------------------------------------------------------------------
def gen_layers(ref, out):
    warnings.filterwarnings("ignore")
    rec = {}
    names = []
    zb = synth.zinc_shape_batch(3, seed=3)
    N, E = zb.num_nodes, zb.num_edges
    ei = torch.from_numpy(zb.edge_index)
    # a second, irregular graph: random multigraph-free digraph with isolated nodes and a hub
    rng = np.random.default_rng(5)
    N2 = 40
    e2 = set()
    while len(e2) < 110:
        a, b = rng.integers(0, N2 - 4, 2)
        if a != b:
            e2.add((int(a), int(b)))
    for t in range(1, 20):
        e2.add((t, 0))
    ei2 = torch.tensor(sorted(e2), dtype=torch.long).T.contiguous()
    ei2 = ei2[:, torch.from_numpy(rng.permutation(ei2.shape[1]))]

    def run(name, cls_name, ctor, graph, d_x, d_id, d_ef, id_scope, train, seed, x_1d=False):
        torch.manual_seed(seed)
        cls = getattr(ref[cls_name], cls_name)
        layer = cls(**ctor)
        # make BN statistics / affine non-trivial
        for m in layer.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.uniform_(-0.5, 0.5)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.uniform_(-0.5, 0.5)
        layer.train(train)
        eidx, n_nodes = graph
        n_edges = eidx.shape[1]
        x = torch.randn(n_nodes, d_x) if not x_1d else torch.randn(n_nodes)
        x.requires_grad_(True)
        kw = {"degrees": torch.randn(n_nodes, ctor.get("d_degree", 1))}
        ids = None
        if d_id is not None:
            ids = torch.randn(n_edges if id_scope == "local" else n_nodes, d_id, requires_grad=True)
            kw["identifiers"] = ids
        elif cls_name == "MPNN_edge_sparse_ogb":
            kw["identifiers"] = None  # the key is read (MPNN_edge_sparse_ogb.py:66) but never used
        ef = None
        if d_ef is not None:
            ef = torch.randn(n_edges, d_ef, requires_grad=True)
            kw["edge_features"] = ef
        sd0 = {k: v.detach().clone() for k, v in layer.state_dict().items()}
        y = layer(x, eidx, **kw)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        rec[name + "/edge_index"] = eidx.numpy()
        rec[name + "/x"] = x.detach().numpy()
        rec[name + "/degrees"] = kw["degrees"].numpy()
        if ids is not None:
            rec[name + "/identifiers"] = ids.detach().numpy()
            rec[name + "/g_identifiers"] = ids.grad.numpy()
        if ef is not None:
            rec[name + "/edge_features"] = ef.detach().numpy()
            rec[name + "/g_edge_features"] = ef.grad.numpy()
        rec[name + "/y"] = y.detach().numpy()
        rec[name + "/w"] = w.numpy()
        rec[name + "/g_x"] = x.grad.numpy()
        for k, v in sd0.items():
            rec[name + "/sd/" + k] = v.numpy()
        for k, p in layer.named_parameters():
            if p.grad is not None:
                rec[name + "/gp/" + k] = p.grad.numpy()
        sd1 = layer.state_dict()
        for k, v in sd1.items():  # BN running stats after a train-mode forward
            if "running" in k or "num_batches" in k:
                rec[name + "/sd_after/" + k] = v.numpy()
        rec[name + "/ctor"] = np.asarray(repr(sorted((k, v) for k, v in ctor.items())))
        rec[name + "/cls"] = np.asarray(cls_name)
        rec[name + "/train"] = np.bool_(train)
        names.append(name)

    graphs = {"zinc": (ei, N), "hub": (ei2, N2)}
    seed = 0
    base = dict(d_degree=1, degree_as_tag=False, retain_features=True, seed=0, aggr="add", eps=0,
                extend_dims=True)
    for gname, graph in graphs.items():
        for msg_kind in ("general", "gin"):
            for id_scope in ("local", "global"):
                for flow in ("source_to_target", "target_to_source"):
                    for bn in (True, False):
                        for train in (False, True):
                            if not bn and train and flow == "target_to_source":
                                continue  # trim
                            act = "relu" if bn else "elu"
                            emb = "one_hot_encoder" if (seed % 2 == 0) else "embedding"
                            # GSN_edge_sparse
                            ctor = dict(base, d_in=7, d_ef=3, d_id=5, id_scope=id_scope, d_msg=12, d_up=10, d_h=[16],
                                        activation_name=act, bn=bn, msg_kind=msg_kind, train_eps=(seed % 3 == 0),
                                        flow=flow, edge_embedding=emb, id_embedding=emb)
                            seed += 1
                            run("GSN_edge_sparse/%s/%s/%s/%s/bn%d/train%d" % (gname, msg_kind, id_scope, flow, bn, train),
                                "GSN_edge_sparse", ctor, graph, 7, 5, 3, id_scope, train, seed)
                            # GSN_sparse
                            ctor = dict(base, d_in=6, d_id=4, id_scope=id_scope, d_msg=None if seed % 5 == 0 else 9,
                                        d_up=11, d_h=[8], activation_name=act, bn=bn, msg_kind=msg_kind,
                                        train_eps=(seed % 3 == 0), flow=flow, id_embedding=emb)
                            seed += 1
                            run("GSN_sparse/%s/%s/%s/%s/bn%d/train%d" % (gname, msg_kind, id_scope, flow, bn, train),
                                "GSN_sparse", ctor, graph, 6, 4, None, id_scope, train, seed)
        for msg_kind in ("general", "gin"):
            for flow in ("source_to_target", "target_to_source"):
                for train in (False, True):
                    ctor = dict(base, d_in=7, d_ef=3, d_msg=12, d_up=10, d_h=[16], activation_name="relu", bn=True,
                                msg_kind=msg_kind, train_eps=False, flow=flow, edge_embedding="one_hot_encoder")
                    seed += 1
                    run("MPNN_edge_sparse/%s/%s/%s/train%d" % (gname, msg_kind, flow, train),
                        "MPNN_edge_sparse", ctor, graph, 7, None, 3, "local", train, seed)
                    ctor = dict(base, d_in=6, d_msg=9, d_up=11, d_h=[8], activation_name="tanh", bn=True,
                                msg_kind=msg_kind, train_eps=True, flow=flow)
                    seed += 1
                    run("MPNN_sparse/%s/%s/%s/train%d" % (gname, msg_kind, flow, train),
                        "MPNN_sparse", ctor, graph, 6, None, None, "local", train, seed)
        for id_scope in ("local", "global"):
            for train in (False, True):
                ctor = dict(base, d_in=8, d_ef=8, d_id=8, id_scope=id_scope, d_msg=None, d_up=8, d_h=[16],
                            activation_name="relu", bn=True, msg_kind="ogb", train_eps=True)
                seed += 1
                run("GSN_edge_sparse_ogb/%s/%s/train%d" % (gname, id_scope, train),
                    "GSN_edge_sparse_ogb", ctor, graph, 8, 8, 8, id_scope, train, seed)
        ctor = dict(base, d_in=8, d_ef=8, d_msg=None, d_up=8, d_h=[16], activation_name="relu", bn=True,
                    msg_kind="ogb", train_eps=False)
        seed += 1
        run("MPNN_edge_sparse_ogb/%s/train0" % gname, "MPNN_edge_sparse_ogb", ctor, graph, 8, None, 8, "local", False, seed)
    # degree_as_tag + 1-D inputs (GSN_sparse.py:96-100)
    ctor = dict(base, d_in=1, d_id=4, id_scope="local", d_msg=9, d_up=11, d_h=[8], activation_name="relu", bn=True,
                msg_kind="general", train_eps=False, flow="source_to_target", id_embedding="one_hot_encoder",
                degree_as_tag=True, retain_features=True, d_degree=1)
    run("GSN_sparse/zinc/degree_as_tag_1d", "GSN_sparse", ctor, graphs["zinc"], 1, 4, None, "local", False, 999, x_1d=True)
    # one ZINC-shaped batch at the real layer-0 widths of config 2 (d_in 28, d_id 12, d_ef 4, d 128), eval + train
    zb2 = synth.zinc_shape_batch(8, seed=4)
    g2 = (torch.from_numpy(zb2.edge_index), zb2.num_nodes)
    for train in (False, True):
        ctor = dict(base, d_in=28, d_ef=4, d_id=12, id_scope="local", d_msg=128, d_up=128, d_h=[128],
                    activation_name="relu", bn=True, msg_kind="general", train_eps=False, flow="source_to_target",
                    edge_embedding="one_hot_encoder", id_embedding="one_hot_encoder")
        run("GSN_edge_sparse/zinc8_real_widths/train%d" % train, "GSN_edge_sparse", ctor, g2, 28, 12, 4, "local", train, 1234)
    rec["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(out, "layers.npz"), **rec)
    print("layers: %d cases" % len(names))


# ----------------------------------------------------------------------------------------------
# preprocessing driver, loaders, encoders
# ----------------------------------------------------------------------------------------------
def _write_tu_fixture(dst_dir):
    """tests/golden/raw/TUTRIM.txt: the first 12 graphs of the reference's IMDBBINARY.txt plus a hand-made graph whose
    adjacency rows name high vertex ids first (exercises the networkx insertion-order edge order) -- data only."""
    src = os.path.join(REF, "datasets/social/IMDBBINARY/IMDBBINARY.txt")
    lines = []
    with open(src) as f:
        f.readline()
        for _ in range(12):
            head = f.readline()
            lines.append(head)
            for _j in range(int(head.split()[0])):
                lines.append(f.readline())
    extra = ["6 1\n", "3 2 4 2\n", "5 1 5\n", "3 2 0 3\n", "7 1 2\n", "3 1 0\n", "5 1 1\n"]
    with open(os.path.join(dst_dir, "TUTRIM.txt"), "w") as f:
        f.write("13\n")
        f.writelines(lines + extra)


def _write_zinc_fixture(dst_dir, rng):
    """tests/golden/raw/ZINC: a synthetic stand-in in the benchmarking-gnns pickle layout (lists of dicts with
    atom_type / dense bond_type / logP_SA_cycle_normalized) + indices/*.index; one molecule has a nonzero diagonal entry
    (a self loop with an edge feature) and one has no bonds at all."""
    import pickle
    os.makedirs(os.path.join(dst_dir, "molecules"), exist_ok=True)
    os.makedirs(os.path.join(dst_dir, "indices"), exist_ok=True)
    for split, n_mol, pick in (("train", 6, [0, 2, 3, 5]), ("val", 3, [1, 2]), ("test", 3, [0, 1])):
        mols = []
        for m in range(n_mol):
            n, ei = synth.zinc_shape_graph(rng)[:2]
            adj = torch.zeros(n, n, dtype=torch.long)
            for (a, b) in ei.T.tolist():
                if a < b:
                    adj[a, b] = adj[b, a] = int(rng.integers(1, 4))
            if split == "train" and m == 2:
                adj[1, 1] = 2
            if split == "val" and m == 2:
                adj.zero_()
            mols.append({"atom_type": torch.from_numpy(rng.integers(0, 28, size=n)), "bond_type": adj,
                         "logP_SA_cycle_normalized": torch.tensor(float(rng.normal()))})
        with open(os.path.join(dst_dir, "molecules", split + ".pickle"), "wb") as f:
            pickle.dump(mols, f)
        with open(os.path.join(dst_dir, "indices", split + ".index"), "w") as f:
            f.write(",".join(str(i) for i in pick) + "\n")


def _dump_prepared(rec, key, graphs):
    rec[key + "/n_graphs"] = np.int64(len(graphs))
    for g, d in enumerate(graphs):
        names = [k_ for k_ in d.__dict__]
        rec["%s/%d/attr_order" % (key, g)] = np.array(names)
        for k_ in names:
            v = getattr(d, k_)
            rec["%s/%d/%s" % (key, g, k_)] = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if isinstance(v, torch.Tensor):
                rec["%s/%d/%s.dtype" % (key, g, k_)] = np.array(str(v.dtype))


def gen_dataset(ref, out):
    import io
    import contextlib
    import shutil
    raw = os.path.join(out, "raw")
    os.makedirs(raw, exist_ok=True)
    rng = np.random.default_rng(11)
    _write_tu_fixture(raw)
    _write_zinc_fixture(os.path.join(raw, "ZINC"), rng)
    shutil.copyfile(os.path.join(REF, "datasets/SR_graphs/sr251256/sr251256.g6"), os.path.join(raw, "sr251256.g6"))
    os.chmod(os.path.join(raw, "sr251256.g6"), 0o644)
    udp, udg, uenc, uu, ugp, uid = (ref[k_] for k_ in ("utils_data_prep", "utils_data_gen", "utils_encoding", "utils",
                                                       "utils_graph_processing", "utils_ids"))
    rec = {}
    sink = io.StringIO()
    # --- loaders
    for tag in (False, True):
        with contextlib.redirect_stdout(sink):
            gl, ncls = udp.load_data(raw, "TUTRIM", tag)
        key = "tu_tag%d" % int(tag)
        rec[key + "/num_classes"] = np.int64(ncls)
        for g, s in enumerate(gl):
            rec["%s/%d/edge_mat" % (key, g)] = s.edge_mat.numpy().reshape(2, -1)
            rec["%s/%d/node_features" % (key, g)] = s.node_features.numpy()
            rec["%s/%d/label" % (key, g)] = np.int64(s.label)
            rec["%s/%d/node_tags" % (key, g)] = np.asarray(s.node_tags, dtype=np.int64)
            rec["%s/%d/max_neighbor" % (key, g)] = np.int64(s.max_neighbor)
    gl, ncls = udp.load_g6_graphs(raw, "sr251256")
    rec["g6/num_classes"] = np.int64(ncls)
    for g, s in enumerate(gl):
        rec["g6/%d/edge_mat" % g] = s.edge_mat.numpy()
        rec["g6/%d/node_features" % g] = s.node_features.numpy()
        rec["g6/%d/label" % g] = s.label.numpy()
    gl, ncls, nnt, net = udp.load_zinc_data(os.path.join(raw, "ZINC"), "ZINC", False)
    rec["zinc/meta"] = np.array([len(gl), ncls, nnt, net], dtype=np.int64)
    for g, s in enumerate(gl):
        rec["zinc/%d/edge_mat" % g] = s.edge_mat.numpy()
        rec["zinc/%d/node_features" % g] = s.node_features.numpy()
        rec["zinc/%d/edge_features" % g] = s.edge_features.numpy()
        rec["zinc/%d/label" % g] = np.float64(float(s.label))

    # --- generate_dataset end to end (utils_data_gen.py:17-81) through the reference's own _prepare / counts2ids
    def run(key, path, name, ks, fam, id_type_fn, count_fn, induced, regression):
        els = [list(fam(k_).edges) for k_ in ks]
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            res = udg.generate_dataset(path, name, ks[-1], uid.subgraph_counts2ids, count_fn, id_type_fn, regression,
                                       "x", multiprocessing=False, num_processes=1, edge_list=els, induced=induced,
                                       directed=False, directed_orbits=False)
        graphs, ncls, nnt, net, sizes = res
        rec[key + "/num_classes"] = np.int64(ncls)
        rec[key + "/orbit_partition_sizes"] = np.asarray(sizes, dtype=np.int64)
        rec[key + "/node_edge_types"] = np.asarray([-1 if nnt is None else nnt, -1 if net is None else net], dtype=np.int64)
        pptr, pflat = pack_edge_lists(els)
        rec[key + "/pattern_ptr"], rec[key + "/pattern_edges"] = pptr, pflat
        _dump_prepared(rec, key, graphs)
        return graphs, sizes

    run("gd_sr25_edge", raw, "sr251256", [3, 4, 5], nx.cycle_graph, ugp.induced_edge_automorphism_orbits,
        ugp.subgraph_isomorphism_edge_counts, True, False)
    tu_graphs, tu_sizes = run("gd_tu_vertex", raw, "TUTRIM", [3, 4], nx.complete_graph, ugp.automorphism_orbits,
                              ugp.subgraph_isomorphism_vertex_counts, False, False)
    run("gd_zinc_vertex", os.path.join(raw, "ZINC"), "ZINC", [3, 4, 5, 6], nx.cycle_graph, ugp.automorphism_orbits,
        ugp.subgraph_isomorphism_vertex_counts, False, True)
    # edge mode on ZINC: the bond-free molecule makes the reference die on an undefined name (utils_data_gen.py:104);
    # record the graphs it does handle by running _prepare per graph
    zg, _, _, _ = udp.load_zinc_data(os.path.join(raw, "ZINC"), "ZINC", False)
    els = [list(nx.cycle_graph(k_).edges) for k_ in (3, 4, 5, 6)]
    dicts = []
    with contextlib.redirect_stdout(sink):
        for el in els:
            sg, part, memb, aut = ugp.induced_edge_automorphism_orbits(edge_list=el, directed=False, directed_orbits=False)
            dicts.append({"subgraph": sg, "orbit_partition": part, "orbit_membership": memb, "aut_count": aut})
    prepared, died = [], []
    for g, s in enumerate(zg):
        try:
            prepared.append(udg._prepare(s, dicts, {"induced": False, "directed": False}, True, "ZINC",
                                         uid.subgraph_counts2ids, ugp.subgraph_isomorphism_edge_counts))
        except NameError:
            died.append(g)
            prepared.append(None)
    assert len(died) == 1
    rec["gd_zinc_edge/reference_nameerror_at"] = np.asarray(died, dtype=np.int64)
    pptr, pflat = pack_edge_lists(els)
    rec["gd_zinc_edge/pattern_ptr"], rec["gd_zinc_edge/pattern_edges"] = pptr, pflat
    rec["gd_zinc_edge/n_graphs"] = np.int64(len(prepared))
    for g, d in enumerate(prepared):
        if d is None:
            continue
        names = [k_ for k_ in d.__dict__]
        rec["gd_zinc_edge/%d/attr_order" % g] = np.array(names)
        for k_ in names:
            v = getattr(d, k_)
            rec["gd_zinc_edge/%d/%s" % (g, k_)] = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)

    # --- downgrade_k (utils.py:332-345): keep the k=3 columns of the k<=4 clique dataset
    dg, dsizes = uu.downgrade_k(tu_graphs, 3, tu_sizes, 3)
    rec["downgrade/sizes"] = np.asarray(dsizes, dtype=np.int64)
    for g, d in enumerate(dg):
        rec["downgrade/%d/identifiers" % g] = d.identifiers.numpy()

    # --- encode (utils_encoding.py:8-59): dataset-level one_hot_unique / one_hot_max of identifiers and degrees
    for enc in ("one_hot_unique", "one_hot_max"):
        with contextlib.redirect_stdout(sink):
            graphs, _ = run("enc_src_" + enc, raw, "sr251256", [3, 4, 5], nx.cycle_graph, ugp.automorphism_orbits,
                            ugp.subgraph_isomorphism_vertex_counts, True, False)
        graphs, enc_ids, d_id, enc_deg, d_deg = uenc.encode(graphs, enc, enc, ids={}, degree={})
        rec["enc_%s/d_id" % enc] = np.asarray(d_id, dtype=np.int64)
        rec["enc_%s/d_degree" % enc] = np.asarray(d_deg, dtype=np.int64)
        for g, d in enumerate(graphs):
            rec["enc_%s/%d/identifiers" % (enc, g)] = np.asarray(d.identifiers.numpy())
            rec["enc_%s/%d/degrees" % (enc, g)] = np.asarray(d.degrees.numpy())
            rec["enc_%s/%d/degrees.dtype" % (enc, g)] = np.array(str(d.degrees.dtype))

    # --- DiscreteEmbedding (utils_graph_learning.py:28-130): one-hot and embedding encoders
    ugl = ref["utils_graph_learning"]
    codes = torch.from_numpy(rng.integers(0, [3, 5, 2, 7], size=(40, 4)))
    d_in = [3, 5, 2, 7]
    rec["emb/codes"] = codes.numpy()
    rec["emb/d_in"] = np.asarray(d_in, dtype=np.int64)
    m = ugl.DiscreteEmbedding("one_hot_encoder", 4, d_in, 16)
    rec["emb/one_hot/out"] = m(codes).numpy()
    rec["emb/one_hot/d_out"] = np.int64(m.d_out)
    for aggr in ("sum", "concat"):
        torch.manual_seed(5)
        with contextlib.redirect_stdout(sink):
            m = ugl.DiscreteEmbedding("embedding", 4, d_in, 16, aggr=aggr, init=None)
        c = codes.clone()
        y = m(c)
        gy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
        (y * gy).sum().backward()
        rec["emb/%s/out" % aggr] = y.detach().numpy()
        rec["emb/%s/gy" % aggr] = gy.numpy()
        rec["emb/%s/d_out" % aggr] = np.int64(m.d_out)
        for k_, v in m.state_dict().items():
            rec["emb/%s/sd/%s" % (aggr, k_)] = v.numpy()
        for k_, p_ in m.named_parameters():
            rec["emb/%s/grad/%s" % (aggr, k_)] = p_.grad.numpy()
    np.savez_compressed(os.path.join(out, "dataset.npz"), **rec)
    print("dataset: %d arrays" % len(rec))


# ----------------------------------------------------------------------------------------------
# whole-model forward: GNNSubstructures (models_graph_classification.py) with the reference's own layers
# ----------------------------------------------------------------------------------------------
def _model_kwargs(model_name, n_layers, d, msg_kind, id_scope, bn, jk_mlp, readout, activation, inject_ids, node_enc,
                  edge_enc, id_emb, d_emb):
    return dict(seed=0, model_name=model_name, readout=readout, dropout_features=[0.0] * (n_layers + 1), bn=[bn] * n_layers,
                final_projection=[True] * (n_layers + 1), inject_ids=inject_ids, inject_edge_features=True,
                random_features=False, id_scope=id_scope, d_msg=[d] * n_layers, d_out=[d] * n_layers, d_h=[[d]] * n_layers,
                aggr="add", flow="source_to_target", msg_kind=msg_kind, train_eps=[False] * n_layers, activation_mlp="relu",
                bn_mlp=True, jk_mlp=jk_mlp, degree_embedding="None", degree_as_tag=[False] * n_layers,
                retain_features=[True] * n_layers, multi_embedding_aggr="sum", input_node_encoder=node_enc,
                d_out_node_encoder=d_emb, edge_encoder=edge_enc, d_out_edge_encoder=[d_emb] * n_layers, id_embedding=id_emb,
                d_out_id_embedding=d_emb, d_out_degree_embedding=d_emb, extend_dims=True, activation=activation)


def gen_model(ref, out):
    import importlib
    import io
    import contextlib
    mgc = importlib.import_module("models_graph_classification")
    assert mgc.__file__.startswith(REF)
    z = np.load(os.path.join(out, "dataset.npz"))
    rec = {}
    sink = io.StringIO()

    def collate(key, n_graphs, id_key="identifiers"):
        xs, eis, ids, efs, batch, degs = [], [], [], [], [], []
        off = 0
        for g in range(n_graphs):
            x = torch.from_numpy(z["%s/%d/x" % (key, g)])
            ei = torch.from_numpy(z["%s/%d/edge_index" % (key, g)])
            xs.append(x); eis.append(ei + off); ids.append(torch.from_numpy(z["%s/%d/%s" % (key, g, id_key)]))
            if "%s/%d/edge_features" % (key, g) in z.files:
                efs.append(torch.from_numpy(z["%s/%d/edge_features" % (key, g)]))
            batch.append(torch.full((x.shape[0],), g, dtype=torch.long))
            degs.append(torch.zeros(x.shape[0]))
            off += x.shape[0]
        d = types.SimpleNamespace(x=torch.cat(xs), edge_index=torch.cat(eis, 1), identifiers=torch.cat(ids),
                                  batch=torch.cat(batch), degrees=torch.cat(degs))
        if efs:
            d.edge_features = torch.cat(efs)
        return d

    def run(name, data, d_id, ctor_args, kw, train):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(sink):
            model = mgc.GNNSubstructures(*ctor_args, **kw)
        # non-trivial BatchNorm statistics and affine parameters
        g = torch.Generator().manual_seed(7)
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
        model.train(train)
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        pred, interm = model(data, return_intermediate=True)
        for k, v in sd0.items():
            rec["%s/sd/%s" % (name, k)] = v.numpy()
        for k, v in model.state_dict().items():
            rec["%s/sd_after/%s" % (name, k)] = v.numpy()
        rec[name + "/pred"] = pred.detach().numpy()
        for i, t in enumerate(interm):
            rec["%s/interm/%d" % (name, i)] = t.detach().numpy()
        if train:
            gy = torch.from_numpy(np.random.default_rng(3).standard_normal(tuple(pred.shape)).astype(np.float32))
            (pred * gy).sum().backward()
            rec[name + "/gy"] = gy.numpy()
            for k, p_ in model.named_parameters():
                if p_.grad is not None:
                    rec["%s/grad/%s" % (name, k)] = p_.grad.numpy()
        for attr in ("x", "edge_index", "identifiers", "batch", "degrees", "edge_features"):
            if hasattr(data, attr):
                rec["%s/data/%s" % (name, attr)] = getattr(data, attr).numpy()
        rec[name + "/d_id"] = np.asarray(d_id, dtype=np.int64)
        rec[name + "/train"] = np.int64(train)

    # (i) README config 1: SR(25,12,5,6), GSN_sparse general/local, one_hot_unique ids, 2 x 64, jk mlp, sum readout, eval
    enc_key = "enc_one_hot_unique"
    n = 15
    data = collate("enc_src_one_hot_unique", n)
    # edge-mode identifiers for the local scope come from the gd_sr25_edge run (induced cycles 3..5)
    data = collate("gd_sr25_edge", n)
    uenc = ref["utils_encoding"]
    ids_list = [torch.from_numpy(z["gd_sr25_edge/%d/identifiers" % g]) for g in range(n)]
    enc = uenc.one_hot_unique(ids_list)
    data.identifiers = torch.cat(enc.fit(ids_list))
    d_id = enc.d
    kw = _model_kwargs("GSN_sparse", 2, 64, "general", "local", True, True, "sum", "relu", False, "None", "None",
                       "one_hot_encoder", 16)
    run("sr25_gsn_sparse_eval", data, d_id, (1, 10, None, d_id), kw, False)
    rec["sr25_gsn_sparse_eval/ctor"] = np.array(["1", "10", "None"])
    kw_t = dict(kw)
    run("sr25_gsn_sparse_train", data, d_id, (1, 10, None, d_id), kw_t, True)

    # (ii) config 2 shape: ZINC fixture, GSN_edge_sparse general/local, one-hot atoms / bonds / ids, 3 x 32, linear jk, mean
    nz = int(z["gd_zinc_edge/n_graphs"])
    keep = [g for g in range(nz) if g not in z["gd_zinc_edge/reference_nameerror_at"].tolist()]
    xs, eis, ids, efs, batch, degs, off = [], [], [], [], [], [], 0
    for j, g in enumerate(keep):
        x = torch.from_numpy(z["gd_zinc_edge/%d/x" % g])
        xs.append(x); eis.append(torch.from_numpy(z["gd_zinc_edge/%d/edge_index" % g]) + off)
        ids.append(torch.from_numpy(z["gd_zinc_edge/%d/identifiers" % g]))
        efs.append(torch.from_numpy(z["gd_zinc_edge/%d/edge_features" % g]))
        batch.append(torch.full((x.shape[0],), j, dtype=torch.long)); degs.append(torch.zeros(x.shape[0]))
        off += x.shape[0]
    enc = uenc.one_hot_unique(ids)
    data = types.SimpleNamespace(x=torch.cat(xs), edge_index=torch.cat(eis, 1), identifiers=torch.cat(enc.fit(ids)),
                                 batch=torch.cat(batch), degrees=torch.cat(degs), edge_features=torch.cat(efs))
    d_id = enc.d
    for train in (False, True):
        kw = _model_kwargs("GSN_edge_sparse", 3, 32, "general", "local", True, False, "mean", "relu", False,
                           "one_hot_encoder", "one_hot_encoder", "one_hot_encoder", 16)
        run("zinc_gsn_edge_%s" % ("train" if train else "eval"), data, d_id, (1, 1, None, d_id, 1, [28], [4]), kw, train)
    # (iii) gin / global with embedding encoders and injected ids (vertex ids of the TU fixture)
    nt = int(z["gd_tu_vertex/n_graphs"])
    data = collate("gd_tu_vertex", nt)
    ids_list = [torch.from_numpy(z["gd_tu_vertex/%d/identifiers" % g]) for g in range(nt)]
    enc = uenc.one_hot_unique(ids_list)
    data.identifiers = torch.cat(enc.fit(ids_list))
    data.x = data.x.argmax(1, keepdim=True)         # node tags as integer codes for the embedding encoder
    d_id = enc.d
    kw = _model_kwargs("GSN_sparse", 2, 32, "gin", "global", True, True, "sum", "elu", True, "embedding", "None",
                       "embedding", 8)
    run("tu_gin_global_eval", data, d_id, (1, 2, None, d_id, None, [int(data.x.max()) + 1]), kw, False)
    # (iv) GNN_OGB (models_graph_classification_ogb_original.py): ogb layers, virtual node, residuals -- config 4 shape
    mogb = importlib.import_module("models_graph_classification_ogb_original")
    assert mogb.__file__.startswith(REF)
    rng = np.random.default_rng(21)
    b = synth.zinc_shape_batch(10, seed=4)
    Nn, Ee = b.num_nodes, b.num_edges
    atom_dims, bond_dims, id_dims = [7, 3, 4], [4, 2], [3, 5]
    dd = types.SimpleNamespace(
        x=torch.from_numpy(rng.integers(0, atom_dims, size=(Nn, 3))), edge_index=torch.from_numpy(b.edge_index),
        edge_features=torch.from_numpy(rng.integers(0, bond_dims, size=(Ee, 2))),
        identifiers=torch.from_numpy(rng.integers(0, id_dims, size=(Ee, 2))),
        batch=torch.from_numpy(np.asarray(b.batch).astype(np.int64)), degrees=torch.zeros(Nn))
    for tag, vn, residual, train in (("ogb_vn_res_eval", True, True, False), ("ogb_vn_train", True, False, True), ("ogb_plain_eval", False, False, False)):
        L, dm = 3, 24
        kw = dict(seed=0, model_name="GSN_edge_sparse_ogb", readout="mean", dropout_features=[0.0] * (L + 1), bn=[True] * L,
                  final_projection=[False] * L + [True], residual=residual, inject_ids=True, vn=vn, id_scope="local",
                  d_msg=[dm] * L, d_out=[dm] * L, d_h=[[2 * dm]] * L, aggr="add", flow="source_to_target", msg_kind="ogb",
                  train_eps=[True] * L, activation_mlp="relu", bn_mlp=True, jk_mlp=False, degree_embedding="None",
                  degree_as_tag=[False] * L, retain_features=[True] * L, multi_embedding_aggr="sum", features_scope="full",
                  input_node_encoder="embedding", d_out_node_encoder=dm, input_vn_encoder="embedding", d_out_vn_encoder=dm,
                  edge_encoder="embedding", d_out_edge_encoder=[dm] * L, id_embedding="embedding", d_out_id_embedding=dm,
                  d_out_degree_embedding=dm, d_out_vn=[dm] * (L - 1), vn_pooling="sum", extend_dims=True, activation="relu")
        torch.manual_seed(0)
        with contextlib.redirect_stdout(sink):
            model = mogb.GNN_OGB(3, 2, None, id_dims, 2, atom_dims, bond_dims, None, None, **kw)
        g = torch.Generator().manual_seed(11)
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
        if vn:      # the reference initialises the virtual node to zero; make it visible
            model.vn_encoder.encoder.encoder[0].weight.data.copy_(torch.randn(1, dm, generator=g) * 0.3)
        model.train(train)
        for k, v in model.state_dict().items():
            rec["%s/sd/%s" % (tag, k)] = v.clone().numpy()
        pred = model(dd)
        for k, v in model.state_dict().items():
            rec["%s/sd_after/%s" % (tag, k)] = v.numpy()
        rec[tag + "/pred"] = pred.detach().numpy()
        rec[tag + "/train"] = np.int64(train)
        rec[tag + "/flags"] = np.asarray([int(vn), int(residual)], dtype=np.int64)
        if train:
            gy = torch.from_numpy(rng.standard_normal(tuple(pred.shape)).astype(np.float32))
            (pred * gy).sum().backward()
            rec[tag + "/gy"] = gy.numpy()
            for k, p_ in model.named_parameters():
                if p_.grad is not None:
                    rec["%s/grad/%s" % (tag, k)] = p_.grad.numpy()
    for attr in ("x", "edge_index", "identifiers", "batch", "degrees", "edge_features"):
        rec["ogb/data/%s" % attr] = getattr(dd, attr).numpy()
    np.savez_compressed(os.path.join(out, "model.npz"), **rec)
    print("model: %d arrays" % len(rec))


def gen_model_ogb300(ref, out):
    """BASELINE config 4 at its REAL widths: models_graph_classification_ogb_original.GNN_OGB, 5 layers x 300, virtual node, one
    train-mode step on 8 molecule-shaped graphs: prediction, BatchNorm running statistics after the step, digests of every parameter
    gradient (tests/helpers.py: grad_digest).  The 3.4 M parameters are not stored: both sides fill the model with
    helpers.procedural_state (a function of the state_dict keys)."""
    import importlib
    import io
    import contextlib
    sys.path.insert(0, os.path.dirname(HERE))
    import helpers
    mogb = importlib.import_module("models_graph_classification_ogb_original")
    assert mogb.__file__.startswith(REF)
    rng = np.random.default_rng(33)
    b = synth.zinc_shape_batch(8, seed=9)
    Nn, Ee = b.num_nodes, b.num_edges
    atom_dims, bond_dims, id_dims = [119, 4, 12, 12, 10, 6, 6, 2, 2], [5, 6, 2], [3, 3, 3, 3]
    dd = types.SimpleNamespace(
        x=torch.from_numpy(rng.integers(0, atom_dims, size=(Nn, len(atom_dims)))), edge_index=torch.from_numpy(b.edge_index),
        edge_features=torch.from_numpy(rng.integers(0, bond_dims, size=(Ee, len(bond_dims)))),
        identifiers=torch.from_numpy(rng.integers(0, id_dims, size=(Ee, len(id_dims)))),
        batch=torch.from_numpy(np.asarray(b.batch).astype(np.int64)), degrees=torch.zeros(Nn))
    L, dm = 5, 300
    kw = dict(seed=0, model_name="GSN_edge_sparse_ogb", readout="mean", dropout_features=[0.0] * (L + 1), bn=[True] * L,
              final_projection=[False] * L + [True], residual=False, inject_ids=True, vn=True, id_scope="local",
              d_msg=[dm] * L, d_out=[dm] * L, d_h=[[2 * dm]] * L, aggr="add", flow="source_to_target", msg_kind="ogb",
              train_eps=[True] * L, activation_mlp="relu", bn_mlp=True, jk_mlp=False, degree_embedding="None",
              degree_as_tag=[False] * L, retain_features=[True] * L, multi_embedding_aggr="sum", features_scope="full",
              input_node_encoder="embedding", d_out_node_encoder=dm, input_vn_encoder="embedding", d_out_vn_encoder=dm,
              edge_encoder="embedding", d_out_edge_encoder=[dm] * L, id_embedding="embedding", d_out_id_embedding=dm,
              d_out_degree_embedding=dm, d_out_vn=[dm] * (L - 1), vn_pooling="sum", extend_dims=True, activation="relu")
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = mogb.GNN_OGB(len(atom_dims), 1, None, id_dims, len(bond_dims), atom_dims, bond_dims, None, None, **kw)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(helpers.procedural_state(shapes))
    model.train(True)
    pred = model(dd)
    rec = {"pred": pred.detach().numpy(), "n_params": np.int64(sum(p.numel() for p in model.parameters()))}
    gy = torch.from_numpy(rng.standard_normal(tuple(pred.shape)).astype(np.float32))
    (pred * gy).sum().backward()
    rec["gy"] = gy.numpy()
    dg = helpers.grad_digest({k: p_.grad for k, p_ in model.named_parameters() if p_.grad is not None})
    rec["grad_keys"] = np.array(sorted(dg))
    rec["grad_digest"] = np.array([dg[k] for k in sorted(dg)], dtype=np.float64)
    after = model.state_dict()
    run_keys = sorted(k for k in after if "running_" in k)
    rec["running_keys"] = np.array(run_keys)
    rec["running_digest"] = np.array([[float(after[k].double().sum()), float(after[k].double().norm())] for k in run_keys])
    rec["shape_keys"] = np.array(sorted(shapes))
    rec["shape_ptr"] = np.cumsum([0] + [len(shapes[k]) for k in sorted(shapes)]).astype(np.int64)
    rec["shape_flat"] = np.array([d for k in sorted(shapes) for d in shapes[k]], dtype=np.int64)
    for attr in ("x", "edge_index", "identifiers", "batch", "degrees", "edge_features"):
        rec["data/%s" % attr] = getattr(dd, attr).numpy()
    np.savez_compressed(os.path.join(out, "model_ogb300.npz"), **rec)
    print("model_ogb300: %d arrays, %d parameters" % (len(rec), int(rec["n_params"])))


def gen_directed(ref, out):
    """directed=True (main.py --directed): patterns and targets are digraphs, vertex counts only (the reference's directed edge
    counter dies on an unbound name, utils_graph_processing.py:146 vs :164).  automorphism_orbits(directed=True) +
    subgraph_isomorphism_vertex_counts(directed=True) of the reference over networkx's DiGraphMatcher."""
    import io
    import contextlib
    ugp = ref["utils_graph_processing"]
    rec = {"names": []}
    pats = [[(0, 1), (1, 2), (2, 0)], [(0, 1), (0, 2), (1, 2)], [(0, 1), (1, 2)], [(0, 1), (1, 0), (1, 2)], [(0, 1), (0, 2), (0, 3)],
            [(1, 0), (2, 0), (3, 0)], [(0, 1), (1, 2), (2, 3), (3, 0)], [(0, 1), (1, 2), (2, 3), (0, 3)], list(nx.cycle_graph(4).edges),
            list(nx.complete_graph(4).edges), [(0, 1), (1, 0), (1, 2), (2, 1)], [(0, 1), (1, 2), (2, 3), (3, 4), (4, 0)],
            [(0, 1), (1, 2), (2, 0), (2, 3), (3, 4)], list(nx.path_graph(5).edges)]
    rng = np.random.default_rng(5)

    def digraph(n, m, loops=0, dups=0):
        src = rng.integers(0, n, size=m); dst = rng.integers(0, n, size=m)
        keep = src != dst
        ei = np.stack([src[keep], dst[keep]]).astype(np.int64)
        if dups:
            ei = np.concatenate([ei, ei[:, :dups]], axis=1)
        if loops:
            lv = rng.integers(0, n, size=loops)
            ei = np.concatenate([ei[:, :3], np.stack([lv, lv]), ei[:, 3:]], axis=1)
        return n, ei
    zb = synth.zinc_shape_batch(4, seed=3)
    graphs = [digraph(10, 30), digraph(16, 70), digraph(24, 110, loops=3, dups=5), digraph(40, 90), zb.graph(0), zb.graph(1),
              (7, np.zeros((2, 0), dtype=np.int64))]
    nn = [10, 16, 24, 43, graphs[4][0], graphs[5][0], 7]
    # orbits of the directed patterns
    for pi, el in enumerate(pats):
        with contextlib.redirect_stdout(io.StringIO()):
            _, part, memb, aut = ugp.automorphism_orbits(edge_list=el, print_msgs=False, directed=True, directed_orbits=False)
        rec["pattern/%d/edges" % pi] = np.asarray(el, dtype=np.int64).reshape(-1, 2)
        rec["pattern/%d/v_membership" % pi] = np.asarray([memb[v] for v in range(len(memb))], dtype=np.int64)
        rec["pattern/%d/aut_count" % pi] = np.int64(aut)
    rec["n_patterns"] = np.int64(len(pats))
    for induced in (False, True):
        dicts = []
        with contextlib.redirect_stdout(io.StringIO()):
            for el in pats:
                sg, part, memb, aut = ugp.automorphism_orbits(edge_list=el, print_msgs=False, directed=True, directed_orbits=False)
                dicts.append({"subgraph": sg, "orbit_partition": part, "orbit_membership": memb, "aut_count": aut})
        outs = []
        for gi, (n, ei) in enumerate(graphs):
            ids = None
            for d in dicts:
                c = ugp.subgraph_isomorphism_vertex_counts(torch.from_numpy(ei), subgraph_dict=d, induced=induced, num_nodes=nn[gi], directed=True)
                ids = c if ids is None else torch.cat((ids, c), 1)
            outs.append(ids.long().numpy())
        name = "digraphs_%s_vertex" % ("induced" if induced else "mono")
        save_case(rec, name, graphs, pats, "vertex", induced, outs, num_nodes=nn)
    rec["names"] = np.asarray(rec["names"])
    np.savez_compressed(os.path.join(out, "counts_directed.npz"), **rec)
    print("directed: %d cases, %d patterns" % (len(rec["names"]), len(pats)))


def gen_cliques(ref, out):
    """The heaviest clique cases of BASELINE config 3, which gen_counts leaves out for networkx-VF2 speed (the three IMDB-BINARY graphs among
    the first 24 with > 1 M K5 maps; K5 on the largest graph: 1.6e9 maps through utils_graph_processing.py:116-127) -- from an enumerator
    that is neither VF2 nor this repo's oracle: networkx.enumerate_all_cliques, tallied per vertex and per directed edge column.
    Reference semantics for K_k, non-induced (one vertex orbit, one edge orbit, |Aut| = k!): counts[v] = # k-cliques containing v
    (utils_graph_processing.py:119-127), counts[(u, v)] = # k-cliques containing u and v (:158-176); the TU loader's edge lists hold every
    pair once per direction, so no duplicate-column rule applies.  The reference's own path is cross-checked here on K3 / K4 of the same
    graphs (VF2 stand-in), so the tally convention is pinned to the reference, not to this function's reading of it."""
    rec = {}
    t0 = time.time()
    clq = lambda ks: [list(nx.complete_graph(k).edges) for k in ks]
    imdb = imdb_graphs()
    sizes = np.array([n for n, _ in imdb])
    heavy = [imdb[i] for i in (8, 11, 15)]
    big = [imdb[int(np.argmax(sizes))]]

    def tally(n, ei, ks):
        g = nx.Graph()
        g.add_nodes_from(range(n))
        g.add_edges_from((int(u), int(v)) for u, v in ei.T if u != v)
        col = {}
        for c in range(ei.shape[1]):
            col[(int(ei[0, c]), int(ei[1, c]))] = c          # (last duplicate wins, utils_graph_processing.py:142-144; none here)
        vc = np.zeros((n, len(ks)), np.int64)
        ec = np.zeros((ei.shape[1], len(ks)), np.int64)
        kmax = max(ks)
        for cl in nx.enumerate_all_cliques(g):              # (by increasing size)
            k = len(cl)
            if k > kmax:
                break
            if k in ks:
                j = ks.index(k)
                for v in cl:
                    vc[v, j] += 1
                for a in cl:
                    for b in cl:
                        if a != b:
                            ec[col[(a, b)], j] += 1
        return vc, ec
    for name, graphs in (("imdb_heavy", heavy), ("imdb_largest", big)):
        ks = [3, 4, 5]
        vs, es = [], []
        for n, ei in graphs:
            vc, ec = tally(n, ei, ks)
            vs.append(vc); es.append(ec)
            print(name, "n=%d E=%d K5 total %d" % (n, ei.shape[1], int(vc[:, 2].sum()) // 5), "%.0fs" % (time.time() - t0), flush=True)
        # the same K3 / K4 columns through the REFERENCE's functions (over the VF2 stand-in): the convention check
        for mode, arr in (("vertex", vs), ("edge", es)):
            o = run_counts(ref, graphs, clq([3, 4]), mode, False)
            for a, b in zip(o, arr):
                assert np.array_equal(a, b[:, :2]), (name, mode)
            save_case(rec, "%s_clique3-5_mono_%s" % (name, mode), graphs, clq(ks), mode, False, arr)
        print(name, "reference K3/K4 cross-check ok", "%.0fs" % (time.time() - t0), flush=True)
    rec["names"] = np.asarray(rec["names"])
    np.savez_compressed(os.path.join(out, "counts_cliques.npz"), **rec)


def gen_stars(ref, out):
    """star_graph(8) -- the 9-vertex pattern of ``--id_type star_graph --k 8`` (utils.py:59-62; GSN_KMAX 8 -> 9 in r06).
    (a) small graphs whose hubs have degree <= 9, through the REFERENCE's functions over the VF2 stand-in (at most 9! / 1! maps per hub),
        vertex and edge mode, induced and not; orbits of the pattern (8! = 40 320 automorphisms) in the file as well;
    (b) the three heaviest IMDB-BINARY graphs (hubs of degree 30-60: 60! / 52! maps -- no enumerator gets there) from the closed form
            centre orbit  C(deg v, 8)                leaf orbit  sum_{u in N(v)} C(deg u - 1, 7)
            edge (u, v)   C(deg u - 1, 7) + C(deg v - 1, 7)        (one undirected edge class: centre-leaf)
        whose convention is pinned to the reference on the graphs of (a) here (asserted equal to the reference's output)."""
    from math import comb
    rec = {}
    t0 = time.time()
    star = list(nx.star_graph(8).edges)
    ugp = ref["utils_graph_processing"]
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        sg, part, memb, aut = ugp.automorphism_orbits(edge_list=star, directed=False, directed_orbits=False)
        sge, eparte, emembe, aute = ugp.induced_edge_automorphism_orbits(edge_list=star, directed=False, directed_orbits=False)
    rec["star8/v_membership"] = np.asarray([memb[v] for v in range(9)], np.int64)
    rec["star8/e_membership"] = np.asarray([emembe[i] for i in range(len(emembe))], np.int64)
    rec["star8/aut_count"] = np.int64(aut)
    rec["star8/edges"] = np.asarray(star, np.int64)
    print("star_graph(8): aut", aut, "vertex orbits", len(part), "edge orbits", len(eparte), "%.0fs" % (time.time() - t0), flush=True)

    def closed(n, ei):
        deg = np.zeros(n, np.int64)
        und = {(int(min(u, v)), int(max(u, v))) for u, v in ei.T if u != v}
        nb = [[] for _ in range(n)]
        for u, v in und:
            deg[u] += 1; deg[v] += 1; nb[u].append(v); nb[v].append(u)
        vc = np.zeros((n, 2), np.int64)
        for v in range(n):
            vc[v, 0] = comb(int(deg[v]), 8)
            vc[v, 1] = sum(comb(int(deg[u]) - 1, 7) for u in nb[v])
        ec = np.zeros((ei.shape[1], 1), np.int64)
        for c in range(ei.shape[1]):
            u, v = int(ei[0, c]), int(ei[1, c])
            if u != v:
                ec[c, 0] = comb(int(deg[u]) - 1, 7) + comb(int(deg[v]) - 1, 7)
        return vc, ec
    rng = np.random.default_rng(88)
    small = []
    for gi in range(6):
        n = int(rng.integers(11, 15))
        und = set()
        hub_deg = 8 + (gi % 2)                                   # one hub of degree 8 or 9, a second vertex of degree 8 in two graphs
        for v in rng.choice(np.arange(1, n), size=hub_deg, replace=False):
            und.add((0, int(v)))
        if gi >= 4:
            for v in rng.choice(np.arange(2, n), size=7, replace=False):
                und.add((1, int(v)))
            und.add((0, 1))
        for _ in range(int(rng.integers(3, 9))):
            a, b = (int(x) for x in rng.choice(np.arange(1, n), size=2, replace=False))
            und.add((min(a, b), max(a, b)))
        # keep every degree <= 9 (the reference enumerates d! / (d - 8)! maps per vertex of degree d)
        deg = np.zeros(n, int)
        keep = []
        for a, b in sorted(und):
            if deg[a] < 9 and deg[b] < 9:
                keep.append((a, b)); deg[a] += 1; deg[b] += 1
        small.append((n, synth.undirected_to_edge_index(n, keep)))
    for mode in ("vertex", "edge"):
        for induced in (False, True):
            o = run_counts(ref, small, [star], mode, induced)
            if not induced:
                for (n, ei), a in zip(small, o):
                    vc, ec = closed(n, ei)
                    assert np.array_equal(a, vc if mode == "vertex" else ec), ("closed form vs reference", mode)
            save_case(rec, "small_star8_%s_%s" % ("ind" if induced else "mono", mode), small, [star], mode, induced, o)
            print("small", mode, induced, "total", int(sum(int(a.sum()) for a in o)), "%.0fs" % (time.time() - t0), flush=True)
    imdb = imdb_graphs()
    heavy = [imdb[i] for i in (8, 11, 15)]
    vs, es = zip(*[closed(n, ei) for n, ei in heavy])
    save_case(rec, "imdb_heavy_star8_mono_vertex", heavy, [star], "vertex", False, list(vs))
    save_case(rec, "imdb_heavy_star8_mono_edge", heavy, [star], "edge", False, list(es))
    print("imdb heavy: centre-orbit total", int(sum(int(v[:, 0].sum()) for v in vs)), flush=True)
    rec["names"] = np.asarray(rec["names"])
    np.savez_compressed(os.path.join(out, "counts_stars.npz"), **rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="orbits,counts,layers")
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    ref = import_reference()
    only = set(args.only.split(","))
    if "orbits" in only:
        gen_orbits(ref, args.out)
    if "counts" in only:
        gen_counts(ref, args.out)
    if "layers" in only:
        gen_layers(ref, args.out)
    if "dataset" in only:
        gen_dataset(ref, args.out)
    if "model" in only:
        gen_model(ref, args.out)
    if "directed" in only:
        gen_directed(ref, args.out)
    if "ogb300" in only:
        gen_model_ogb300(ref, args.out)
    if "stars" in only:
        gen_stars(ref, args.out)
    if "cliques" in only:
        gen_cliques(ref, args.out)


if __name__ == "__main__":
    main()

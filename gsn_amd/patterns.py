"""Pattern (substructure) analysis: host-side mirror of the reference's orbit functions.

Same names, argument meaning and return tuples as ``utils_graph_processing.py`` of the reference
(automorphism_orbits :10-56, induced_edge_automorphism_orbits :58-100, edge_automorphism_orbits :189-251) so that
``utils.process_arguments`` / ``utils_data_gen.generate_dataset`` can call them unchanged.  The work is done by
``gsn_pattern_orbits`` in libgsn_hip.so (host C++, no graph-tool).  The first element of every returned tuple -- a
``gt.Graph`` in the reference, opaque to its callers -- is a small picklable
:class:`PatternGraph` here.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _abi

__all__ = ["PatternGraph", "automorphism_orbits", "induced_edge_automorphism_orbits", "edge_automorphism_orbits",
           "analyse"]


class PatternGraph:
    """Stand-in for the ``gt.Graph`` stored under ``subgraph_dict['subgraph']``: the pattern's edge list plus what
    the counting kernel needs to rebuild its plan (``directed_orbits``).  Picklable (joblib workers)."""

    def __init__(self, edge_list, directed_orbits=False, line_graph_orbits=False, directed=False):
        self.edge_list = [(int(u), int(v)) for u, v in edge_list]
        self.directed_orbits = bool(directed_orbits)
        self.directed = bool(directed)            # gt.Graph(directed=True): the rows of edge_list are arcs
        # produced by edge_automorphism_orbits (the deprecated line-graph variant): its membership dict has one entry per
        # undirected edge, which the reference's edge counter indexes by directed-edge position (see counting.py)
        self.line_graph_orbits = bool(line_graph_orbits)

    def get_edges(self):
        """Simple undirected edges (directed: arcs) in insertion order, like ``gt.Graph.get_edges()`` after
        remove_self_loops / remove_parallel_edges."""
        seen, out = set(), []
        for u, v in self.edge_list:
            key = (u, v) if getattr(self, "directed", False) else (min(u, v), max(u, v))
            if u != v and key not in seen:
                seen.add(key)
                out.append([u, v])
        return np.asarray(out, dtype=np.int64).reshape(-1, 2)

    def get_vertices(self):
        return np.arange(self.num_vertices())

    def num_vertices(self):
        return max(max(u, v) for u, v in self.edge_list) + 1

    def key(self):
        return tuple(self.edge_list)

    def __repr__(self):
        return "PatternGraph(%r)" % (self.edge_list,)


def analyse(edge_list, directed_orbits=False, directed=False):
    """-> dict(k, vertex_orbit[k], n_vertex_orbits, arcs[2m,2], arc_orbit[2m], n_edge_orbits, aut_count).
    ``directed``: the rows of edge_list are arcs (the reference's ``directed=True``, utils_graph_processing.py:14-16)."""
    e = np.ascontiguousarray(np.asarray(list(edge_list), dtype=np.int64).reshape(-1, 2))
    L = _abi.lib()
    k, nvo, na, neo, aut = (ctypes.c_int64() for _ in range(5))
    vorb = np.zeros(16, dtype=np.int64)          # (GSN_KMAX = 9 vertices, at most 72 arcs)
    arcs = np.zeros((128, 2), dtype=np.int64)
    aorb = np.zeros(128, dtype=np.int64)
    rc = L.gsn_pattern_orbits(len(e), _abi.ptr(e), int(bool(directed_orbits)) | (2 if directed else 0), ctypes.addressof(k), _abi.ptr(vorb),
                              ctypes.addressof(nvo), _abi.ptr(arcs), _abi.ptr(aorb), ctypes.addressof(na),
                              ctypes.addressof(neo), ctypes.addressof(aut))
    _abi.check(rc, "gsn_pattern_orbits")
    return dict(k=int(k.value), vertex_orbit=vorb[:k.value].copy(), n_vertex_orbits=int(nvo.value),
                arcs=arcs[:na.value].copy(), arc_orbit=aorb[:na.value].copy(), n_edge_orbits=int(neo.value),
                aut_count=int(aut.value))


def automorphism_orbits(edge_list, print_msgs=True, **kwargs):
    """Vertex automorphism orbits.  Returns ``(graph, orbit_partition, orbit_membership, aut_count)`` exactly as
    utils_graph_processing.py:10-56: ``orbit_membership[v]`` = rank of the smallest vertex of v's orbit,
    ``orbit_partition[orbit]`` = vertices in ascending order.  ``directed=True`` (:14): the rows of ``edge_list`` are
    arcs and the automorphisms are those of the digraph."""
    directed = bool(kwargs.get("directed", False))
    info = analyse(edge_list, False, directed)
    orbit_membership = {v: int(info["vertex_orbit"][v]) for v in range(info["k"])}
    orbit_partition = {}
    for v, o in orbit_membership.items():
        orbit_partition.setdefault(o, []).append(v)
    if print_msgs:
        print("Orbit partition of given substructure: {}".format(orbit_partition))
        print("Number of orbits: {}".format(len(orbit_partition)))
        print("Automorphism count: {}".format(info["aut_count"]))
    return PatternGraph(edge_list, False, directed=directed), orbit_partition, orbit_membership, info["aut_count"]


def induced_edge_automorphism_orbits(edge_list, **kwargs):
    """Edge orbits induced by the vertex orbits.  Returns ``(graph, edge_orbit_partition, edge_orbit_membership,
    aut_count)`` as utils_graph_processing.py:58-100: membership is indexed by position in the pattern's sorted
    bidirectional edge list; orbit ids in first-seen order of {orbit(u), orbit(v)} (ordered iff directed_orbits)."""
    if kwargs.get("directed", False):
        raise NotImplementedError("directed edge orbits feed the reference's directed edge counter, which fails on an unbound "
                                  "name (utils_graph_processing.py:146 vs :164); directed patterns are vertex-count only")
    directed_orbits = bool(kwargs.get("directed_orbits", False))
    info = analyse(edge_list, directed_orbits)
    edge_orbit_partition, edge_orbit_membership = {}, {}
    for i, (arc, o) in enumerate(zip(info["arcs"].tolist(), info["arc_orbit"].tolist())):
        edge_orbit_partition.setdefault(int(o), []).append(tuple(arc))
        edge_orbit_membership[i] = int(o)
    print("Edge orbit partition of given substructure: {}".format(edge_orbit_partition))
    print("Number of edge orbits: {}".format(len(edge_orbit_partition)))
    print("Graph (vertex) automorphism count: {}".format(info["aut_count"]))
    return PatternGraph(edge_list, directed_orbits), edge_orbit_partition, edge_orbit_membership, info["aut_count"]


def graph_vertex_orbits(n_vertices, edges):
    """Vertex orbits of Aut(G) for a graph with <= 64 vertices -> (orbit id per vertex, number of orbits); orbit id =
    rank of the orbit's smallest vertex (gsn_graph_vertex_orbits, host C++)."""
    e = np.ascontiguousarray(np.asarray(list(edges), dtype=np.int64).reshape(-1, 2))
    orb = np.zeros(max(int(n_vertices), 1), dtype=np.int64)
    n_orb = ctypes.c_int64()
    rc = _abi.lib().gsn_graph_vertex_orbits(int(n_vertices), len(e), _abi.ptr(e) if len(e) else None, _abi.ptr(orb),
                                            ctypes.addressof(n_orb))
    _abi.check(rc, "gsn_graph_vertex_orbits")
    return orb[:int(n_vertices)].copy(), int(n_orb.value)


def edge_automorphism_orbits(edge_list, **kwargs):
    """Edge orbits from the vertex orbits of the pattern's LINE graph -- the reference's deprecated
    ``--edge_automorphism line_graph`` variant (utils_graph_processing.py:189-251).  Returns ``(graph, orbit_partition,
    orbit_membership, aut_count)``: ``orbit_partition[orbit]`` lists the pattern edges as the line graph's node tuples,
    ``orbit_membership[i]`` is indexed by position in ``graph.get_edges()`` (:241-243), ``aut_count`` = |Aut(H)| of the
    pattern itself (:201-202).  The line graph and its node order come from networkx, the reference's own dependency for
    this step (:193, :206-210); graph-tool's automorphism enumeration of it (:214-224) is replaced by
    ``gsn_graph_vertex_orbits``.  Like the reference, a line-graph node that touches no line-graph edge (a pattern
    component with a single edge) is not a vertex of the graph whose orbits are taken, and raises KeyError."""
    import networkx as nx
    if kwargs.get("directed", False):
        raise NotImplementedError("directed patterns are not supported")
    info = analyse(edge_list, False)
    graph = PatternGraph(edge_list, False, line_graph_orbits=True)
    line = nx.line_graph(nx.from_edgelist(edge_list))
    mapping = {node: i for i, node in enumerate(line.nodes)}
    inverse_mapping = {i: node for node, i in mapping.items()}
    line_edges = [(mapping[a], mapping[b]) for a, b in line.edges]
    n_line = max((max(a, b) for a, b in line_edges), default=-1) + 1     # gt.Graph.add_edge_list: vertices 0..max id
    orbit, _ = graph_vertex_orbits(n_line, line_edges)
    orbit_membership = {v: int(orbit[v]) for v in range(n_line)}
    orbit_partition = {}
    for vertex, orb in orbit_membership.items():
        orbit_partition.setdefault(orb, []).append(inverse_mapping[vertex])
    orbit_membership_new = {}
    for i, edge in enumerate(graph.get_edges().tolist()):
        edge = tuple(edge)
        mapped_edge = mapping[edge] if edge in mapping else mapping[(edge[1], edge[0])]
        orbit_membership_new[i] = orbit_membership[mapped_edge]
    print("Edge orbit partition of given substructure: {}".format(orbit_partition))
    print("Number of edge orbits: {}".format(len(orbit_partition)))
    print("Graph (vertex) automorphism count: {}".format(info["aut_count"]))
    return graph, orbit_partition, orbit_membership_new, info["aut_count"]

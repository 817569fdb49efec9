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

    def __init__(self, edge_list, directed_orbits=False):
        self.edge_list = [(int(u), int(v)) for u, v in edge_list]
        self.directed_orbits = bool(directed_orbits)

    def get_edges(self):
        """Simple undirected edges in insertion order, like ``gt.Graph.get_edges()`` after
        remove_self_loops / remove_parallel_edges."""
        seen, out = set(), []
        for u, v in self.edge_list:
            key = (min(u, v), max(u, v))
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


def analyse(edge_list, directed_orbits=False):
    """-> dict(k, vertex_orbit[k], n_vertex_orbits, arcs[2m,2], arc_orbit[2m], n_edge_orbits, aut_count)."""
    e = np.ascontiguousarray(np.asarray(list(edge_list), dtype=np.int64).reshape(-1, 2))
    L = _abi.lib()
    k, nvo, na, neo, aut = (ctypes.c_int64() for _ in range(5))
    vorb = np.zeros(8, dtype=np.int64)
    arcs = np.zeros((64, 2), dtype=np.int64)
    aorb = np.zeros(64, dtype=np.int64)
    rc = L.gsn_pattern_orbits(len(e), _abi.ptr(e), int(bool(directed_orbits)), ctypes.addressof(k), _abi.ptr(vorb),
                              ctypes.addressof(nvo), _abi.ptr(arcs), _abi.ptr(aorb), ctypes.addressof(na),
                              ctypes.addressof(neo), ctypes.addressof(aut))
    _abi.check(rc, "gsn_pattern_orbits")
    return dict(k=int(k.value), vertex_orbit=vorb[:k.value].copy(), n_vertex_orbits=int(nvo.value),
                arcs=arcs[:na.value].copy(), arc_orbit=aorb[:na.value].copy(), n_edge_orbits=int(neo.value),
                aut_count=int(aut.value))


def automorphism_orbits(edge_list, print_msgs=True, **kwargs):
    """Vertex automorphism orbits.  Returns ``(graph, orbit_partition, orbit_membership, aut_count)`` exactly as
    utils_graph_processing.py:10-56: ``orbit_membership[v]`` = rank of the smallest vertex of v's orbit,
    ``orbit_partition[orbit]`` = vertices in ascending order."""
    if kwargs.get("directed", False):
        raise NotImplementedError("directed patterns are not supported (the reference's directed edge path is broken "
                                  "too: utils_graph_processing.py:146 vs :164)")
    info = analyse(edge_list, False)
    orbit_membership = {v: int(info["vertex_orbit"][v]) for v in range(info["k"])}
    orbit_partition = {}
    for v, o in orbit_membership.items():
        orbit_partition.setdefault(o, []).append(v)
    if print_msgs:
        print("Orbit partition of given substructure: {}".format(orbit_partition))
        print("Number of orbits: {}".format(len(orbit_partition)))
        print("Automorphism count: {}".format(info["aut_count"]))
    return PatternGraph(edge_list, False), orbit_partition, orbit_membership, info["aut_count"]


def induced_edge_automorphism_orbits(edge_list, **kwargs):
    """Edge orbits induced by the vertex orbits.  Returns ``(graph, edge_orbit_partition, edge_orbit_membership,
    aut_count)`` as utils_graph_processing.py:58-100: membership is indexed by position in the pattern's sorted
    bidirectional edge list; orbit ids in first-seen order of {orbit(u), orbit(v)} (ordered iff directed_orbits)."""
    if kwargs.get("directed", False):
        raise NotImplementedError("directed patterns are not supported")
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


def edge_automorphism_orbits(edge_list, **kwargs):
    """Deprecated in the reference itself (utils_graph_processing.py:185 "line graph edge automorphism: deprecated",
    only reachable with --edge_automorphism line_graph, used by no README / BASELINE config).  Kept for the import in
    utils.py:2; not implemented on the HIP path."""
    raise NotImplementedError("edge_automorphism='line_graph' (deprecated in the reference) is not implemented; "
                              "use the default --edge_automorphism induced")

"""Shared helpers for the parity tests: golden-fixture access."""
import ast
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_CACHE = {}


def load(name):
    if name not in _CACHE:
        _CACHE[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return _CACHE[name]


def case_names(name):
    return [str(s) for s in load(name)["names"]]


def pattern_lists(z, case):
    ptr = z[case + "/pattern_ptr"]
    flat = z[case + "/pattern_edges"]
    return [flat[ptr[i]:ptr[i + 1]].tolist() for i in range(len(ptr) - 1)]


def count_case(case, fixture="counts"):
    z = load(fixture)
    return dict(
        node_ptr=z[case + "/node_ptr"], edge_ptr=z[case + "/edge_ptr"], edge_index_local=z[case + "/edge_index_local"],
        patterns=pattern_lists(z, case), mode=str(z[case + "/mode"]), induced=bool(z[case + "/induced"]),
        directed_orbits=bool(z[case + "/directed_orbits"]), counts=z[case + "/counts"])


def directed_patterns():
    """[(edge list, vertex membership, aut_count)] of the digraph patterns in counts_directed.npz (reference:
    automorphism_orbits(directed=True))."""
    z = load("counts_directed")
    return [(z["pattern/%d/edges" % i].tolist(), z["pattern/%d/v_membership" % i].tolist(), int(z["pattern/%d/aut_count" % i]))
            for i in range(int(z["n_patterns"]))]


def layer_case(case):
    z = load("layers")
    pre = case + "/"
    d = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    d["ctor"] = dict(ast.literal_eval(str(d["ctor"])))
    d["cls"] = str(d["cls"])
    d["train"] = bool(d["train"])
    return d

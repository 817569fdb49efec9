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


def procedural_state(shapes, seed=0):
    """A deterministic state_dict from key names and shapes alone (for models too large to commit their parameters: the 5 x 300
    virtual-node model of BASELINE config 4 has 3.4 M of them).  The generator script fills the REFERENCE model with it, the test
    fills ours with it: only outputs and digests are stored.  2-D weights ~ randn / sqrt(fan_in), embeddings randn * 0.3, BatchNorm
    affine weights in [0.5, 1.5], running_var in [0.5, 1.5], running_mean and biases small, integer buffers zero."""
    import zlib
    import torch
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        g = torch.Generator().manual_seed((zlib.crc32(k.encode()) + 7919 * seed) % (2 ** 31))
        if k.endswith("num_batches_tracked"):
            t = torch.zeros(shp, dtype=torch.long)
        elif k.endswith("running_var") or (k.endswith("weight") and len(shp) == 1):
            t = torch.rand(shp, generator=g) + 0.5
        elif k.endswith("running_mean") or k.endswith("bias"):
            t = torch.randn(shp, generator=g) * 0.1
        elif len(shp) == 2 and "encoder" in k:
            t = torch.randn(shp, generator=g) * 0.3
        elif len(shp) == 2:
            t = torch.randn(shp, generator=g) / float(shp[1]) ** 0.5
        else:
            t = torch.randn(shp, generator=g) * 0.1
        out[k] = t
    return out


def grad_digest(named_grads, seed=1):
    """Per parameter: (projection of the gradient on a fixed random direction made from the key, its 2-norm, its largest magnitude) --
    what the 5 x 300 model's golden stores instead of 3.4 M gradient entries."""
    import zlib
    import torch
    out = {}
    for k, gr in named_grads.items():
        g = torch.Generator().manual_seed((zlib.crc32(k.encode()) + 104729 * seed) % (2 ** 31))
        r = torch.randn(tuple(gr.shape), generator=g)
        g64 = gr.detach().double().cpu()
        out[k] = (float((g64 * r.double()).sum()), float(g64.norm()), float(g64.abs().max()))
    return out

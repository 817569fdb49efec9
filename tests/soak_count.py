#!/usr/bin/env python3
"""Soak run of the counting kernel against the C oracle with fresh seeds (test infrastructure, not collected by pytest): random
undirected graphs (Erdos-Renyi of several densities, with repeated pairs now and then, 1-300 vertices so that all
bit-matrix widths and the multi-workgroup split occur), random sets of connected patterns with 3-6 vertices, vertex and edge mode,
induced and not, symmetric and one-directional edge lists.  Integer work: every count must be identical.

    python tests/soak_count.py [first_seed] [n_seeds]"""
import os
import sys

import networkx as nx
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsn_amd import synth  # noqa: E402
from gsn_amd.counting import counts2ids_batch  # noqa: E402
from oracle import oracle  # noqa: E402

ATLAS = [g for g in nx.graph_atlas_g() if 3 <= g.number_of_nodes() <= 6 and g.number_of_edges() > 0 and nx.is_connected(g)]


def graph(rng):
    n = int(rng.choice([1, 2, 5, 12, 23, 40, 64, 65, 100, 129, 200, 300], p=[.02, .03, .1, .15, .2, .15, .08, .07, .08, .05, .04, .03]))
    dens = float(rng.choice([0.5, 1.0, 1.5, 3.0])) if n > 2 else 1.0
    m = int(min(n * dens, n * (n - 1) / 2)) if n > 1 else 0
    g = nx.gnm_random_graph(n, m, seed=int(rng.integers(1 << 30)))
    e = np.array(g.edges, dtype=np.int64).reshape(-1, 2)
    ei = np.concatenate([e, e[:, ::-1]], 0).T if len(e) else np.zeros((2, 0), np.int64)
    if ei.shape[1] and rng.random() < 0.2:                      # repeated pairs, a self loop
        k = int(rng.integers(1, 4))
        ei = np.concatenate([ei, ei[:, :k], ei[::-1, :k]], 1)   # (self loops are stripped by the caller, utils_ids.py:12)
    if ei.shape[1]:
        ei = ei[:, rng.permutation(ei.shape[1])]
    return n, ei


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    fails = cases = 0
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        graphs = [graph(rng) for _ in range(int(rng.integers(1, 40)))]
        pats = [list(ATLAS[int(i)].edges) for i in rng.choice(len(ATLAS), size=int(rng.integers(1, 6)), replace=False)]
        mode = "edge" if rng.random() < 0.5 else "vertex"
        induced = bool(rng.random() < 0.5)
        b = synth.collate(graphs)
        local = b.edge_index - np.repeat(b.node_ptr[:-1], np.diff(b.edge_ptr))[None, :]
        ref = oracle.counts2ids(mode, induced, b.node_ptr, b.edge_ptr, local, pats, n_threads=16)
        got = counts2ids_batch(b, pats, mode, induced).cpu().numpy()
        cases += 1
        if got.shape != ref.shape or not np.array_equal(got, ref):
            fails += 1
            bad = np.argwhere(got != ref)[:4].tolist() if got.shape == ref.shape else "shape"
            print("FAIL seed %d: %d graphs, %s, induced %s, %d patterns: %s" % (seed, len(graphs), mode, induced, len(pats), bad), flush=True)
    print("count soak: %d batches, %d failures" % (cases, fails))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())

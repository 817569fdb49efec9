#!/usr/bin/env python3
"""Soak run of the counting launch's side outputs (gsn_count_encode_pack16_side_hip; test infrastructure, not collected by pytest): random
collated batches -- graphs of 0 .. 300 vertices (every bit-matrix width), empty graphs, repeated pairs, self loops, columns in random order --
vertex and edge mode, both CSR rows, one to three code columns.  Against definitions, not against other kernels: the CSR = numpy's stable
argsort by target, the packs = one-hot of the codes, the identifiers = a plain counting launch.  Launch configurations that refuse the side
outputs (a graph split over workgroups) must refuse them loudly.

    python tests/soak_side.py [first_seed] [n_seeds]         (PYTORCH_NO_CUDA_MEMORY_CACHING=1: any access past a tensor faults)"""
import os
import sys

import networkx as nx
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsn_amd import _abi, layers, synth  # noqa: E402
from gsn_amd.counting import CountPlan, count_batch, count_batch_side  # noqa: E402


def graph(rng, sizes, probs):
    n = int(rng.choice(sizes, p=probs))
    m = int(min(n * float(rng.choice([0.5, 1.0, 2.0])), n * (n - 1) / 2)) if n > 1 else 0
    g = nx.gnm_random_graph(n, m, seed=int(rng.integers(1 << 30)))
    e = np.array(g.edges, dtype=np.int64).reshape(-1, 2)
    ei = np.concatenate([e, e[:, ::-1]], 0).T if len(e) else np.zeros((2, 0), np.int64)
    if ei.shape[1] and rng.random() < 0.3:
        k = int(rng.integers(1, 4))
        loops = np.repeat(rng.integers(0, n, (1, k)), 2, 0)
        ei = np.concatenate([ei, ei[:, :k], ei[::-1, :k], loops], 1)       # repeated pairs (both directions), self loops
    if ei.shape[1]:
        ei = ei[:, rng.permutation(ei.shape[1])]
    return n, ei


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda", 0)
    fails = refused = cases = 0
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        kind = int(rng.integers(3))
        if kind == 0:       # molecules and smaller (one-word rows; pairs of graphs per workgroup)
            sizes, probs, ng = [0, 1, 2, 9, 23, 38, 64], [.05, .05, .1, .2, .3, .2, .1], int(rng.integers(1, 400))
        elif kind == 1:     # mixed widths, enough graphs for one workgroup per graph
            sizes, probs, ng = [1, 30, 65, 100, 128], [.1, .3, .2, .2, .2], int(rng.integers(2048, 2300))
        else:               # few large graphs: the launch splits them -> refusal
            sizes, probs, ng = [100, 200, 300], [.4, .3, .3], int(rng.integers(1, 12))
        graphs = [graph(rng, sizes, probs) for _ in range(ng)]
        b = synth.collate(graphs)
        N, E = b.num_nodes, b.num_edges
        mode = "edge" if rng.random() < 0.5 else "vertex"
        pats = [list(nx.cycle_graph(3).edges)] + ([list(nx.path_graph(3).edges)] if rng.random() < 0.5 else [])
        plan = CountPlan.get(pats, mode, False)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        node_ptr, edge_ptr, ei = t(b.node_ptr), t(b.edge_ptr), t(b.edge_index)
        mn, me = max(int(np.diff(b.node_ptr).max()), 1), int(np.diff(b.edge_ptr).max())
        ncols = int(rng.integers(1, 4))
        ncls = [int(c) for c in rng.integers(1, 9, ncols)]
        codes = np.stack([rng.integers(0, c + (1 if rng.random() < 0.2 else 0), N) for c in ncls], 1).astype(np.int64) if N else np.zeros((0, ncols), np.int64)
        clamp = bool(rng.random() < 0.5)
        xc = layers.Codes(t(codes), ncls, clamp=clamp, check=False)
        row = int(rng.integers(2))
        cases += 1
        try:
            r = count_batch_side(plan, node_ptr, edge_ptr, ei, mn, me, x_codes=xc, csr_row=row, register=False)
        except _abi.GsnError as ex:
            if "side workgroups" in str(ex) or "do not sort in LDS" in str(ex):
                refused += 1
                continue
            raise
        torch.cuda.synchronize()
        bad = []
        ids_ref, _ = count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, check=False)
        st = r["status"].cpu().numpy()
        ok_rows = np.repeat(st == 0, np.diff(b.edge_ptr if mode == "edge" else b.node_ptr))
        if not np.array_equal(r["ids"].cpu().numpy()[ok_rows], ids_ref.cpu().numpy()[ok_rows]):
            bad.append("identifiers")
        order = np.argsort(b.edge_index[row], kind="stable")
        c = r["csr"]
        seg = np.concatenate([[0], np.cumsum(np.bincount(b.edge_index[row], minlength=N))]) if E else np.zeros(N + 1, np.int64)
        if not (np.array_equal(c.perm.cpu().numpy(), order) and np.array_equal(c.seg_ptr.cpu().numpy(), seg)
                and np.array_equal(c.tgt.cpu().numpy(), b.edge_index[row][order]) and np.array_equal(c.src.cpu().numpy(), b.edge_index[1 - row][order])):
            bad.append("csr")
        want = np.zeros((N, 32), np.float32)
        off = 0
        any_out = False
        for j, cc in enumerate(ncls):
            v = codes[:, j].copy()
            if clamp:
                v = np.clip(v, 0, cc - 1)
            inr = (v >= 0) & (v < cc)
            any_out = any_out or bool((~inr).any())
            want[np.nonzero(inr)[0], off + v[inr]] = 1.0
            off += cc
        want[:, 31] = 1.0
        if not np.array_equal(r["node_pack"].float().cpu().numpy(), want):
            bad.append("node pack")
        if int(r["code_status"].item()) != int(any_out):
            bad.append("code status %d vs %d" % (int(r["code_status"].item()), int(any_out)))
        if bad:
            fails += 1
            print("FAIL seed %d (kind %d, %d graphs, %s, row %d): %s" % (seed, kind, ng, mode, row, ", ".join(bad)), flush=True)
    print("side soak: %d batches, %d refused (split launches), %d failures" % (cases, refused, fails))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""star_graph patterns (a GSN family: utils.py subgraph families) on clique-union ego networks (IMDB-like: hubs of degree 30-60): the run-of-twins
closed form C(n, r) against the search it replaces (GSN_COUNT_PLAIN_TAILS is not a knob: compare with the numbers in DESIGN.md)."""
import os, sys, time
import numpy as np, torch, networkx as nx
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from gsn_amd import synth
from gsn_amd.counting import CountPlan, count_batch
import bench_configs
rng = np.random.default_rng(0)
graphs = [bench_configs.clique_union_graph(rng, int(rng.integers(12, 60)), int(rng.integers(1, 4)), int(rng.integers(6, 14))) for _ in range(1000)]
b = synth.collate(graphs)
dev = torch.device("cuda", 0)
node_ptr, edge_ptr = torch.from_numpy(b.node_ptr).to(dev), torch.from_numpy(b.edge_ptr).to(dev)
ei = torch.from_numpy(b.edge_index).to(dev)
for ks in ([3, 4, 5], [3, 4, 5, 6, 7], [8]):
    pats = [list(nx.star_graph(k - 1).edges) for k in ks]
    plan = CountPlan.get(pats, "vertex", False)
    out = torch.empty((b.num_nodes, plan.n_cols), dtype=torch.int64, device=dev)
    f = lambda: count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=int(np.diff(b.node_ptr).max()), max_edges=int(np.diff(b.edge_ptr).max()), device=dev, out=out, check=False)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); f(); f(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("star_graph k in %s on 1000 ego nets: %.3f ms, column sums %s" % (ks, dt * 1e3, out.sum(0).tolist()[:6]))

"""r06: does the order of the graphs matter for the molecule instantiation of the counting kernel (two graphs per wave)?  The bench's 65 536-graph
ZINC-shaped batch, rings 3..6 in edge mode, int64 identifiers: as they come, sorted by vertex count (neighbours in a wave are alike), by falling
edge count."""
import os
import sys
import time

import networkx as nx
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsn_amd import synth  # noqa: E402
from gsn_amd.counting import CountPlan, count_batch  # noqa: E402

dev = torch.device("cuda", 0)
G = 65536
b = synth.zinc_shape_batch(G, seed=1000)
plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", False)
node_ptr, edge_ptr, ei = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (b.node_ptr, b.edge_ptr, b.edge_index))
mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
out = torch.empty((b.num_edges, 4), dtype=torch.int64, device=dev)
nn, ne = np.diff(b.node_ptr), np.diff(b.edge_ptr)
ref = None
for name, ids in (("as they come", None), ("by vertex count", np.argsort(nn, kind="stable")), ("by falling edge count", np.argsort(-ne, kind="stable")),
                  ("by falling vertex count", np.argsort(-nn, kind="stable"))):
    gi = None if ids is None else torch.from_numpy(ids.astype(np.int32)).to(dev)
    f = lambda: count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, out=out, check=False, graph_ids=gi)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    if ref is None:
        ref = out.clone()
    print("%-24s %7.4f ms   same counts: %s" % (name, dt * 1e3, bool(torch.equal(out, ref))), flush=True)

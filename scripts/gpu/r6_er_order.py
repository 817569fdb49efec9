"""r06: does the ORDER of the graphs inside a counting launch matter for BASELINE config 5 (one workgroup per graph, 2 048 graphs on 1 536 slots)?
Graphs handed to the kernel by falling estimated cost (sum_v deg^4: dist.counting_cost), by rising cost, and as they come."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsn_amd import dist, synth  # noqa: E402
from gsn_amd.counting import CountPlan, count_batch  # noqa: E402

dev = torch.device("cuda", 0)
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "orbits.npz"))
pats = [z["all_simple_graphs_5/%d/edges" % i].tolist() for i in range(21)]
plan = CountPlan.get(pats, "vertex", False)
G = int(os.environ.get("G", "2048"))
b = synth.collate([synth.er_graph(128, 1000, s) for s in range(G)])
cost = np.asarray(dist.counting_cost(b.edge_index, b.edge_ptr, 5), dtype=np.float64)
print("cost spread: min %.3g  median %.3g  max %.3g" % (cost.min(), np.median(cost), cost.max()))
node_ptr, edge_ptr, ei = (torch.from_numpy(a).to(dev) for a in (b.node_ptr, b.edge_ptr, b.edge_index))
out = torch.empty((b.num_nodes, plan.n_cols), dtype=torch.int64, device=dev)
me = int(np.diff(b.edge_ptr).max())
ref = None
for name, ids in (("as they come", None), ("falling cost", np.argsort(-cost, kind="stable")), ("rising cost", np.argsort(cost, kind="stable"))):
    f = lambda: count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=128, max_edges=me, device=dev, out=out, check=False,
                            graph_ids=None if ids is None else ids.astype(np.int32))
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    if ref is None:
        ref = out.clone()
    print("%-14s %8.2f ms  %8.0f graphs/s   same counts: %s" % (name, dt * 1e3, G / dt, bool(torch.equal(out, ref))), flush=True)

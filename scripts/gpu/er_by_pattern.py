#!/usr/bin/env python3
"""ER G(128,1000) x 1024: counting time per five-vertex pattern (vertex mode, non-induced) and which of its plans end in a closed form."""
import os, sys, time
import numpy as np, torch, networkx as nx
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gsn_amd import synth
from gsn_amd.counting import CountPlan, count_batch
z = np.load(os.path.join(ROOT, "tests", "golden", "orbits.npz"))
pats = [z["all_simple_graphs_5/%d/edges" % i].tolist() for i in range(21)]
dev = torch.device("cuda", 0)
b = synth.collate([synth.er_graph(128, 1000, s) for s in range(1024)])
node_ptr, edge_ptr = torch.from_numpy(b.node_ptr).to(dev), torch.from_numpy(b.edge_ptr).to(dev)
ei = torch.from_numpy(b.edge_index).to(dev)
tot = 0.0
for i, p in enumerate(pats):
    plan = CountPlan.get([p], "vertex", False)
    arr = None
    for name in dir(plan):
        v = getattr(plan, name)
        if isinstance(v, np.ndarray) and v.dtype == np.uint32:
            arr = v; break
    n_plans, plans_off = int(arr[3]), int(arr[7])
    tails = [(int(arr[plans_off + j * 12 + 1]) >> 28) & 3 for j in range(n_plans)]
    out = torch.empty((b.num_nodes, plan.n_cols), dtype=torch.int64, device=dev)
    f = lambda: count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=128, max_edges=int(np.diff(b.edge_ptr).max()), device=dev, out=out, check=False)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); f(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    tot += dt
    g = nx.Graph(p)
    print("pattern %2d edges %d degs %s plans %d tails %s  %.2f ms" % (i, g.number_of_edges(), sorted(dict(g.degree()).values()), n_plans, tails, dt * 1e3))
print("sum %.2f ms" % (tot * 1e3))

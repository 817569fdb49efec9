import sys, torch, networkx as nx
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsn_amd import synth
from gsn_amd.counting import CountPlan, count_batch
dev = torch.device('cuda', 0)
b = synth.zinc_shape_batch(65536, seed=5)
node_ptr, edge_ptr, ei = (torch.from_numpy(a).to(dev) for a in (b.node_ptr, b.edge_ptr, b.edge_index))
mn, me = int(max(b.node_ptr[1:] - b.node_ptr[:-1])), int(max(b.edge_ptr[1:] - b.edge_ptr[:-1]))
def t(ks, mode, induced=False):
    plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in ks], mode, induced)
    rows = b.num_edges if mode == 'edge' else b.num_nodes
    out = torch.empty((rows, plan.n_cols), dtype=torch.int64, device=dev)
    f = lambda: count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, check=False, out=out)
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    print(mode, list(ks), 'plans', plan.n_plans, 'ms %.4f' % (e0.elapsed_time(e1) / 30), 'sum', int(out.sum()))
for ks in ([3], [4], [5], [6], [3, 4], [3, 4, 5], [3, 4, 5, 6], [3, 4, 5, 6, 7, 8]):
    t(ks, 'edge')
for ks in ([3], [6], [3, 4, 5, 6]):
    t(ks, 'vertex')

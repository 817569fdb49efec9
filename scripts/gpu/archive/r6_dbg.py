import numpy as np, torch, networkx as nx, sys
sys.path.insert(0,'/root/repo')
from gsn_amd import layers
from gsn_amd.counting import CountPlan, count_batch, count_batch_side
rng = np.random.default_rng(5)
node_ptr, edge_ptr, cols = [0], [0], []
for g in range(41):
    n = int(rng.integers(2, 30)); m = int(rng.integers(0, 90))
    u = rng.integers(0, n, m); v = rng.integers(0, n, m)
    cols.append(np.stack([u, v]) + node_ptr[-1])
    node_ptr.append(node_ptr[-1] + n); edge_ptr.append(edge_ptr[-1] + m)
ei_np = np.concatenate(cols, 1).astype(np.int64)
dev = torch.device('cuda',0)
node_ptr_t, edge_ptr_t, ei = (torch.tensor(a, dtype=torch.int64, device=dev) for a in (node_ptr, edge_ptr, ei_np))
mn, me = int(np.diff(node_ptr).max()), int(np.diff(edge_ptr).max())
plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3,6)], "edge", False)
ref1,st1 = count_batch(plan, node_ptr_t, edge_ptr_t, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, check=False)
ref2,st2 = count_batch(plan, node_ptr_t, edge_ptr_t, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, check=False)
print("ref deterministic:", torch.equal(ref1,ref2), "statuses", st1.cpu().numpy())
for row in (0,1):
    r = count_batch_side(plan, node_ptr_t, edge_ptr_t, ei, mn, me, csr_row=row, register=False)
    d = (r["ids"] != ref1).any(1).nonzero().flatten().cpu().numpy()
    gid = np.searchsorted(np.asarray(edge_ptr), d, side='right') - 1
    print("row", row, "diff rows", len(d), "graphs", np.unique(gid), "status", r["status"].cpu().numpy()[np.unique(gid)] if len(d) else None)
    if len(d):
        print(r["ids"][d[:5]].cpu().numpy(), ref1[d[:5]].cpu().numpy())
r = count_batch_side(plan, node_ptr_t, edge_ptr_t, ei, mn, me, csr_row=None, register=False)
print("no csr side: equal", torch.equal(r["ids"], ref1))

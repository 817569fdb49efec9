"""The counting kernel of the headline step with and without the int64 rows (counts=False: the encoded rows / pack only)."""
import os, sys, numpy as np, torch, networkx as nx
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from gsn_amd import packs
from gsn_amd.counting import CountPlan, count_batch
dev = torch.device("cuda", 0)
b = bench.make_batch(65536, seed=1000)
E = b.num_edges
node_ptr, edge_ptr = torch.from_numpy(b.node_ptr).to(dev), torch.from_numpy(b.edge_ptr).to(dev)
ei = torch.from_numpy(b.edge_index).to(dev)
mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", False)
ids = torch.empty((E, plan.n_cols), dtype=torch.int64, device=dev)
enc = torch.empty((E, 12), dtype=torch.float32, device=dev)
epack = packs.new_edge_pack(E, dev)
def run(counts, pack):
    return count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, check=False, encode=([3, 3, 3, 3], True),
                       counts=counts, out=ids if counts else None, encoded_out=enc, encoded_pack=(epack, 0) if pack else None)
os.environ["GSN_CHAIN_TRACE"] = "1"
run(False, True); run(True, True)
os.environ.pop("GSN_CHAIN_TRACE")
for rep in range(2):
    for counts in (False, True):
        for pack in (True, False):
            for _ in range(20): run(counts, pack)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run(counts, pack)
            e1.record(); torch.cuda.synchronize()
            print("int64 rows %s  pack %s: %.4f ms" % (counts, pack, e0.elapsed_time(e1) / 50))

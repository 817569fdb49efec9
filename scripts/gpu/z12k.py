"""The 12 000-graph step of bench.py (BASELINE configs[1] dataset size) three ways: eager wall clock (host enqueue vs GPU), HIP events around the step,
one captured HIP graph replayed -- which of them the wall clock of the eager loop measures."""
import os, sys, time
import networkx as nx
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from gsn_amd import flags, layers, packs
from gsn_amd.counting import CountPlan, count_batch
from gsn_amd.graphs import GraphedStep

dev = torch.device("cuda", 0)
b2 = bench.make_batch(int(os.environ.get("G", "12000")), seed=77)
plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", True)
torch.manual_seed(0)
layer = layers.GSN_edge_sparse(d_in=28, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=128, d_up=128,
                               d_h=[128], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target").to(dev).eval()
sel = layer._sel()
np2, ep2 = torch.from_numpy(b2.node_ptr).to(dev), torch.from_numpy(b2.edge_ptr).to(dev)
ei2 = torch.from_numpy(b2.edge_index).to(dev)
deg2 = torch.zeros(b2.num_nodes, device=dev)
ids2 = torch.empty((b2.num_edges, plan.n_cols), dtype=torch.int64, device=dev)
mn2, me2 = int(np.diff(b2.node_ptr).max()), int(np.diff(b2.edge_ptr).max())
layers.set_graph_partition(ei2, np2, ep2, mn2, me2, check=False)
xc2 = layers.Codes(torch.from_numpy(b2.atom_type).to(dev), [28]); efc2 = layers.Codes(torch.from_numpy(b2.bond_type).to(dev), [4])
npk2, epk2 = packs.new_node_pack(b2.num_nodes, dev), packs.new_edge_pack(b2.num_edges, dev)
flags.CODE_STATUS_CHECK = False
side = torch.cuda.Stream(device=dev, priority=-1)

def step(fork=True):
    layers._CSR_CACHE.clear()
    main = torch.cuda.current_stream(dev)
    def indep():
        layers._csr_for(ei2, sel, b2.num_nodes); packs.pack_node_codes(xc2, npk2); packs.pack_edge_codes(efc2, epk2, 12)
    if fork:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            indep()
    else:
        indep()
    idc = count_batch(plan, np2, ep2, ei2, ids_are_global=True, max_nodes=mn2, max_edges=me2, device=dev, check=False, encode=([3, 3, 3, 3], True),
                      counts=True, out=ids2, encoded_pack=(epk2, 0), encoded_rows=False)[2]
    if fork:
        main.wait_stream(side)
    with torch.no_grad():
        return layer(xc2, ei2, identifiers=idc, degrees=deg2, edge_features=efc2)

for fork in (True, False):
    for _ in range(300): step(fork)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): step(fork)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): step(fork)
    e1.record(); torch.cuda.synchronize()
    print("fork", fork, "eager: host enqueue %.4f ms/step, wall %.4f ms/step, HIP events %.4f ms/step" % (t_enq / 300 * 1e3, t_all / 300 * 1e3, e0.elapsed_time(e1) / 300))
g = GraphedStep(lambda: step(False), warmup=3, device=dev)
for _ in range(100): g()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): g()
torch.cuda.synchronize()
print("hip graph (one chain): %.4f ms/step" % ((time.perf_counter() - t0) / 300 * 1e3))

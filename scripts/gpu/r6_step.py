"""r06: the one-call step (gsn_amd.step.CountLayerStep: counting with side outputs + layer 0) against the six-launch composition of r05, on the
bench's 65 536-graph ZINC-shaped batch: device time per step, host enqueue time per step, the kernels one by one."""
import os
import sys
import time

import networkx as nx
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsn_amd import flags, layers, packs, synth  # noqa: E402
from gsn_amd.counting import CountPlan, count_batch, count_batch_side  # noqa: E402
from gsn_amd.step import CountLayerStep  # noqa: E402

G = int(os.environ.get("G", "65536"))
K = int(os.environ.get("K", "20"))
dev = torch.device("cuda", 0)
b = synth.zinc_shape_batch(G, seed=1000)
N, E = b.num_nodes, b.num_edges
mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
node_ptr, edge_ptr, ei, atoms, bonds = t(b.node_ptr), t(b.edge_ptr), t(b.edge_index), t(b.atom_type), t(b.bond_type)
plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", False)
CTOR = dict(d_in=28, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=128,
            d_up=128, d_h=[128], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target")
torch.manual_seed(0)
layer = layers.GSN_edge_sparse(**CTOR).to(dev).eval()
xc, efc = layers.Codes(atoms, [28]), layers.Codes(bonds, [4])
flags.CODE_STATUS_CHECK = False
ids_out = torch.empty((E, 4), dtype=torch.int64, device=dev)
y_out = torch.empty((N, 128), dtype=torch.float32, device=dev)
step = CountLayerStep(plan, layer, [3, 3, 3, 3])
degrees = torch.zeros(N, device=dev)
layers.set_graph_partition(ei, node_ptr, edge_ptr, mn, me, check=False)
epack_c, npack_c = packs.new_edge_pack(E, dev), packs.new_node_pack(N, dev)


def old_step():
    layers._CSR_CACHE.clear()
    layers._csr_for(ei, 1, N)
    packs.pack_node_codes(xc, npack_c)
    packs.pack_edge_codes(efc, epack_c, 12)
    idc = count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, check=False, encode=([3, 3, 3, 3], True),
                      counts=True, out=ids_out, encoded_pack=(epack_c, 0), encoded_rows=False)[2]
    with torch.no_grad():
        return layer(xc, ei, identifiers=idc, degrees=degrees, edge_features=efc)


def new_step():
    return step(node_ptr, edge_ptr, ei, xc, efc, mn, me, ids_out=ids_out, out=y_out)[1]


def side_only():
    count_batch_side(plan, node_ptr, edge_ptr, ei, mn, me, id_classes=[3, 3, 3, 3], x_codes=xc, ef_codes=efc, csr_row=1, out=ids_out, register=False)


def count_only():
    count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, check=False, encode=([3, 3, 3, 3], True),
                counts=True, out=ids_out, encoded_pack=(epack_c, 0), encoded_rows=False)


def timeit(fn, name):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-28s %.4f ms per step   host enqueue %.4f ms per step" % (name, dt / K * 1e3, t_enq / K * 1e3), flush=True)
    return dt / K * 1e3


y_old = old_step().clone()
ids_old = ids_out.clone()
y_new = new_step().clone()
print("equal rows:", bool(torch.equal(y_old, y_new)), "equal ids:", bool(torch.equal(ids_old, ids_out)))
for rep in range(2):
    timeit(old_step, "six-launch composition")
    timeit(new_step, "one-call step")
    timeit(count_only, "count (ids + pack cols)")
    timeit(side_only, "count + side outputs")
for name, kw in (("side: csr only", dict(csr_row=1)), ("side: node pack only", dict(x_codes=xc)), ("side: ids pack only", dict(id_classes=[3, 3, 3, 3])),
                 ("side: ids pack + bond codes", dict(id_classes=[3, 3, 3, 3], ef_codes=efc)), ("side: csr + node pack", dict(csr_row=1, x_codes=xc)),
                 ("side: nothing", dict()), ("side: csr + ids pack", dict(csr_row=1, id_classes=[3, 3, 3, 3]))):
    timeit(lambda: count_batch_side(plan, node_ptr, edge_ptr, ei, mn, me, out=ids_out, register=False, n_nodes=N, **kw), name)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(device=dev)
s.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(s):
    new_step()
torch.cuda.current_stream(dev).wait_stream(s)
with torch.cuda.graph(g):
    new_step()
timeit(g.replay, "one-call step, HIP graph")

# (bench.py's capture allocates the layer rows INSIDE the capture: from the graph's private pool)
def new_step_fresh():
    return step(node_ptr, edge_ptr, ei, xc, efc, mn, me, ids_out=ids_out)[1]
timeit(new_step_fresh, "one-call step, fresh rows")
g2 = torch.cuda.CUDAGraph()
s.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(s):
    new_step_fresh()
torch.cuda.current_stream(dev).wait_stream(s)
with torch.cuda.graph(g2):
    y_g2 = new_step_fresh()
timeit(g2.replay, "HIP graph, rows from its pool")
timeit(g.replay, "HIP graph, caller's rows")
timeit(new_step, "one-call step")

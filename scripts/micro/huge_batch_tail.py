#!/usr/bin/env python3
"""The headline step (count + encode + GSN-e layer 0) on 1 M graphs (24 M vertices, 50 M edge rows, 3.1e9 output elements): the
outputs of the LAST 4096 graphs must equal those of the same graphs run as a batch of their own (offsets past 2^31 elements, the
last tiles of every kernel).  Counts and encoded rows bit-exact, layer rows to 1e-5."""
import os, sys, time
import networkx as nx, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gsn_amd import layers, synth
from gsn_amd.counting import CountPlan, count_batch
dev = torch.device("cuda")
G, T = int(os.environ.get("G", 1 << 20)), 4096
t0 = time.time(); b = synth.zinc_shape_batch(G, seed=77); print("batch built in %.0f s" % (time.time() - t0), flush=True)
plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", False)
layer = layers.GSN_edge_sparse(**bench.CTOR).to(dev).eval()
for m in layer.modules():
    if isinstance(m, torch.nn.BatchNorm1d):
        m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.5, 1.5)
mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
def run(g0, g1):
    n0, n1, e0, e1 = int(b.node_ptr[g0]), int(b.node_ptr[g1]), int(b.edge_ptr[g0]), int(b.edge_ptr[g1])
    node_ptr = torch.from_numpy(b.node_ptr[g0:g1 + 1] - n0).to(dev); edge_ptr = torch.from_numpy(b.edge_ptr[g0:g1 + 1] - e0).to(dev)
    ei = torch.from_numpy(b.edge_index[:, e0:e1] - n0).to(dev)
    cnt, st, enc = count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, encode=([3, 3, 3, 3], True))
    assert int(st.max()) == 0
    x = torch.nn.functional.one_hot(torch.from_numpy(b.atom_type[n0:n1]).to(dev), 28).float()
    ef = torch.nn.functional.one_hot(torch.from_numpy(b.bond_type[e0:e1]).to(dev), 4).float()
    layers.set_graph_partition(ei, node_ptr, edge_ptr, mn, me)
    with torch.no_grad():
        y = layer(x, ei, identifiers=enc, degrees=torch.zeros(n1 - n0, device=dev), edge_features=ef)
    return cnt, enc, y
cnt, enc, y = run(0, G)
print("big batch: N %d E %d, output elements %.2e" % (y.shape[0], enc.shape[0], y.numel()), flush=True)
nt, et = int(b.node_ptr[G] - b.node_ptr[G - T]), int(b.edge_ptr[G] - b.edge_ptr[G - T])
cnt_t, enc_t, y_t = cnt[-et:].clone(), enc[-et:].clone(), y[-nt:].clone()
del cnt, enc, y; torch.cuda.empty_cache()
c2, e2, y2 = run(G - T, G)
print("tail of the big batch vs its own batch: counts equal %s, encoded rows equal %s, layer max err / max %.2e" % (
    bool(torch.equal(cnt_t, c2)), bool(torch.equal(enc_t, e2)), float((y_t - y2).abs().max() / y2.abs().max())))

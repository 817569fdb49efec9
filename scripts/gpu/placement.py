"""Does the layer kernel's time depend on WHERE its output lands?  One big buffer, the output as a view at chosen byte offsets."""
import os, sys, numpy as np, torch, networkx as nx
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from gsn_amd import layers, packs
dev = torch.device("cuda", 0)
b = bench.make_batch(65536, seed=1000)
N, E = b.num_nodes, b.num_edges
ei = torch.from_numpy(b.edge_index).to(dev)
x = torch.nn.functional.one_hot(torch.from_numpy(b.atom_type), 28).float().to(dev)
ef = torch.nn.functional.one_hot(torch.from_numpy(b.bond_type), 4).float().to(dev)
ids = torch.nn.functional.one_hot(torch.randint(0, 3, (E, 4), generator=torch.Generator().manual_seed(2)), 3).reshape(E, 12).float().to(dev)
deg = torch.zeros(N, device=dev)
torch.manual_seed(0)
layer = layers.GSN_edge_sparse(**bench.CTOR).to(dev).eval()
packs.node_pack(x); packs.edge_pack([ids, ef])
big = torch.empty(4 * 1024 ** 3, dtype=torch.uint8, device=dev)
real_empty = torch.empty
state = {"off": None}
def fake_empty(*a, **k):
    shape = a[0] if a and isinstance(a[0], (tuple, list, torch.Size)) else a
    if state["off"] is not None and tuple(shape) == (N, 128) and k.get("dtype") is torch.float32:
        return big[state["off"]:state["off"] + N * 128 * 4].view(torch.float32).view(N, 128)
    return real_empty(*a, **k)
torch.empty = fake_empty
print("big base %x" % big.data_ptr(), "x %x" % x.data_ptr())
with torch.no_grad():
    for _ in range(20): layer(x, ei, identifiers=ids, degrees=deg, edge_features=ef)
    for off in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 3 << 20, 16 << 20, (1 << 30), (1 << 30) + (1 << 20) + 8192, None, 0, None):
        state["off"] = off
        for _ in range(5): layer(x, ei, identifiers=ids, degrees=deg, edge_features=ef)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): y = layer(x, ei, identifiers=ids, degrees=deg, edge_features=ef)
        e1.record(); torch.cuda.synchronize()
        print("offset %s: %.4f ms  (out at %x)" % (off, e0.elapsed_time(e1) / 30, y.data_ptr()))

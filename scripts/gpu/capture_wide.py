"""A d = 128 layer forward inside a captured HIP graph: the d = 128 kernel with caller-owned scratch is captured."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gsn_amd import layers, synth
b = synth.zinc_shape_batch(64, seed=3)
N, E = b.num_nodes, b.num_edges
dev = "cuda"
torch.manual_seed(0)
ctor = dict(d_in=128, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=128, d_up=128,
            d_h=[128], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target")
layer = layers.GSN_edge_sparse(**ctor).to(dev).eval()
x = torch.randn(N, 128, device=dev); ids = torch.randn(E, 12, device=dev); ef = torch.randn(E, 4, device=dev)
ei = torch.from_numpy(b.edge_index).to(dev); deg = torch.zeros(N, device=dev)
with torch.no_grad():
    y0 = layer(x, ei, identifiers=ids, degrees=deg, edge_features=ef)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            layer(x, ei, identifiers=ids, degrees=deg, edge_features=ef)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    os.environ["GSN_CHAIN_TRACE"] = "1"
    with torch.cuda.graph(g):
        y1 = layer(x, ei, identifiers=ids, degrees=deg, edge_features=ef)
    os.environ.pop("GSN_CHAIN_TRACE")
    g.replay()
    torch.cuda.synchronize()
print("capture ok, max diff vs eager", float((y0 - y1).abs().max() / y0.abs().max()))

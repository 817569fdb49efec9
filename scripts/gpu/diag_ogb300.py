import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
import test_big_batch_gpu as T
from gsn_amd import layers, synth
from oracle import oracle
cls, d, kind = "MPNN_edge_sparse_ogb", 300, "ogb"
for ng in (256, 1024):
    torch.manual_seed(d + len(cls) + 1)
    b = synth.zinc_shape_batch(ng, seed=22)
    N, E = b.num_nodes, b.num_edges
    ctor, d_x, d_id, d_ef = T._case(cls, d, kind)
    layer = getattr(layers, cls)(**ctor); layer.train()
    x = torch.randn(N, d_x); ei = torch.from_numpy(b.edge_index); ef = torch.randn(E, d_ef) * 0.5
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in layer.state_dict().items()}
    xr = x.clone().requires_grad_(True); efr = ef.clone().requires_grad_(True)
    ref = oracle.layer_forward(cls, ctor, sd, xr, ei, identifiers=None, degrees=None, edge_features=efr, training=True)
    w_out = torch.randn_like(ref); (ref * w_out).sum().backward()
    # fp64 reference of the same
    sd64 = {k: v.detach().double().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in layer.state_dict().items()}
    x64 = x.double().requires_grad_(True); ef64 = ef.double().requires_grad_(True)
    try:
        ref64 = oracle.layer_forward(cls, ctor, sd64, x64, ei, identifiers=None, degrees=None, edge_features=ef64, training=True)
        (ref64 * w_out.double()).sum().backward(); have64 = True
    except Exception as e:
        print("fp64 oracle failed:", str(e)[:100]); have64 = False
    layer.cuda()
    xg = x.cuda().requires_grad_(True); efg = ef.cuda().requires_grad_(True)
    y = layer(xg, ei.cuda(), degrees=torch.zeros(N, device="cuda"), identifiers=None, edge_features=efg)
    (y * w_out.cuda()).sum().backward()
    sc = float(xr.grad.abs().max())
    e32 = (xg.grad.cpu() - xr.grad).abs() / sc
    print(ng, "fwd rel", float((y.detach().cpu() - ref.detach()).abs().max() / ref.abs().max()), "dx: max", float(e32.max()), "n>3e-4", int((e32 > 3e-4).sum()),
          "cols with >3e-4:", int(((e32 > 3e-4).sum(0) > 0).sum()), "rows:", int(((e32 > 3e-4).sum(1) > 0).sum()), "median", float(e32.median()))
    if have64:
        eo = (xr.grad.double() - x64.grad).abs() / sc; eg = (xg.grad.cpu().double() - x64.grad).abs() / sc
        print("   vs fp64: oracle32 max", float(eo.max()), "n>3e-4", int((eo > 3e-4).sum()), "| ours max", float(eg.max()), "n>3e-4", int((eg > 3e-4).sum()))
    # the per-edge blocks' gradient (d relu(x_j + e) / d e): the same kinks, seen from the edge rows
    sce = float(efr.grad.abs().max())
    ee = (efg.grad.cpu() - efr.grad).abs() / sce
    print("   d ef: max", float(ee.max()), "rows with an entry >= 2e-5:", int(((ee >= 2e-5).sum(1) > 0).sum()), "of", E, "median", float(ee.median()),
          "| GSN_LINEAR_F16X3_STATS", os.environ.get("GSN_LINEAR_F16X3_STATS", "default"))
    if have64:
        eo = (efr.grad.double() - ef64.grad).abs() / sce; eg = (efg.grad.cpu().double() - ef64.grad).abs() / sce
        print("   d ef vs fp64: oracle32 max", float(eo.max()), "rows", int(((eo >= 2e-5).sum(1) > 0).sum()), "| ours max", float(eg.max()), "rows", int(((eg >= 2e-5).sum(1) > 0).sum()))

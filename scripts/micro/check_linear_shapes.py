"""fp16x3 linear path against fp64 at shapes with odd / even K slice counts, one or many row tiles per workgroup, one or several
column tiles: prints the rows and columns with an error above 1e-4 of the largest output (none expected)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsn_amd import layers
dev = torch.device("cuda")
shapes = ((196608, 600, 300), (4096, 600, 300), (4096, 608, 300), (4096, 600, 384), (20000, 352, 300), (20000, 600, 256), (196608, 300, 600),
          (380317, 260, 128), (100000, 128, 128), (5000, 36, 64), (300, 260, 100), (70000, 132, 96))
for (M, K, N) in shapes:
    torch.manual_seed(0)
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    st = layers._Stage(W, b, None, "relu", [(x, None)])
    y = layers._launch_stages([st], M)
    ref = torch.relu(x.double() @ W.double().T + b.double())
    d = (y.double() - ref).abs()
    bad = d > 1e-4 * ref.abs().max()
    rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
    print((M, K, N), "max err %.2e" % float(d.max() / ref.abs().max()), "bad rows", rows.numel(), rows[:6].tolist(), "bad cols", cols.numel(), cols[:6].tolist())

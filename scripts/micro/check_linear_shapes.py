"""fp16x3 linear path against fp64 at shapes with odd / even K slice counts and one or many row tiles per workgroup: prints the rows and
columns with an error above 1e-4 of the largest output (none expected)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gsn_amd import layers
dev = torch.device("cuda")
for (M, K, N) in ((196608, 600, 300), (4096, 600, 300), (4096, 608, 300), (4096, 600, 128*3), (20000, 352, 300), (20000, 600, 256), (196608, 300, 600)):
    torch.manual_seed(0)
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    st = layers._Stage(W, b, None, "relu", [(x, None)])
    y = layers._launch_stages([st], M)
    ref = torch.relu(x.double() @ W.double().T + b.double())
    d = (y.double() - ref).abs()
    bad = d > 1e-4 * ref.abs().max()
    rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
    print((M, K, N), "max err", float(d.max() / ref.abs().max()), "bad rows", rows.numel(), rows[:6].tolist(), rows[-3:].tolist(), "bad cols", cols.numel(), cols[:6].tolist(), cols[-3:].tolist())

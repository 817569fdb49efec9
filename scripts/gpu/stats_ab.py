"""Train-mode dense stage (pre-BN rows + fp64 column sums) on the bf16x6 kernel vs the fp16x3 kernel (+ statistics), both against fp64:
row error relative to the row's largest magnitude, statistics error relative to the column's own scale."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gsn_amd import flags, layers

dev = "cuda:0"
torch.manual_seed(0)
for (M, K, N, kind) in ((5000, 300, 600, "randn"), (5000, 600, 300, "relu"), (256, 300, 600, "randn"), (105083, 300, 600, "relu"), (105083, 600, 300, "relu")):
    x = torch.randn(M, K, device=dev)
    if kind == "relu":
        x = torch.relu(x) * torch.rand(M, 1, device=dev) * 4
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev) * 0.1
    ref = x.double() @ w.double().t() + b.double()
    rs, rq = ref.sum(0), (ref * ref).sum(0)
    for name, f16 in (("bf16x6", False), ("fp16x3", True)):
        flags.LINEAR_F16X3_STATS = f16
        flags.LINEAR_F16X3_MIN_TILES = 0 if f16 else 10 ** 9
        stats = torch.zeros(2 * N, dtype=torch.float64, device=dev)
        y = layers._linear_hip([(x, None)], w, b, None, None, None, 0, M, out=True, stats=stats)
        torch.cuda.synchronize()
        e = (y.double() - ref).abs()
        rowmax = ref.abs().amax(1, keepdim=True)
        es = (stats[:N] - rs).abs() / (ref.abs().sum(0))
        eq = (stats[N:] - rq).abs() / rq
        print("%-7s M %6d K %3d N %3d %-5s rows: max err/rowmax %.3g mean %.3g | sums: max %.3g  squares: max %.3g" %
              (name, M, K, N, kind, float((e / rowmax).max()), float((e / rowmax).mean()), float(es.max()), float(eq.max())), flush=True)

#!/usr/bin/env python3
"""Dense stages with more than 2^31 output elements (32-bit offset overflow check): the last rows against fp64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsn_amd import layers
dev = torch.device("cuda")
for (M, K, N, how) in ((5_000_000, 64, 600, "linear"), (20_000_000, 32, 128, "linear"), (20_000_000, 32, 128, "chain"), (18_000_000, 20, 120, "chain2")):
    torch.manual_seed(0)
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    if how == "linear":
        y = layers._linear_hip([(x, None)], W, b, None, None, None, 1, M)
        ref = lambda xs: torch.relu(xs.double() @ W.double().T + b.double())
    elif how == "chain":
        y = layers._launch_stages([layers._Stage(W, b, None, "relu", [(x, None)])], M)
        ref = lambda xs: torch.relu(xs.double() @ W.double().T + b.double())
    else:
        W2 = torch.randn(N, N, device=dev) / N ** 0.5
        y = layers._launch_stages([layers._Stage(W, b, None, "relu", [(x, None)]), layers._Stage(W2, None, None, "identity", [])], M)
        ref = lambda xs: torch.relu(xs.double() @ W.double().T + b.double()) @ W2.double().T
    worst = 0.0
    for lo in (0, M // 2 - 1000, M - 4096):
        r = ref(x[lo:lo + 4096]); d = (y[lo:lo + 4096].double() - r).abs().max() / r.abs().max()
        worst = max(worst, float(d))
    print((M, K, N, how), "elements %.2e" % (M * N), "max rel err over sampled rows %.2e" % worst, flush=True)
    del x, y
    torch.cuda.empty_cache()

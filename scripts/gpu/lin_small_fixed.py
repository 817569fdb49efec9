"""Fixed cost of the 32-row-tile dense kernel: the same product 50 times inside a replayed HIP graph at K = 32 .. 1216 (plain / column statistics /
a transposed weight view)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gsn_amd import flags, layers
flags.LINEAR_F16X3 = False
dev = torch.device("cuda", 0)
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def graphed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(torch.cuda.Stream()):
        with torch.cuda.graph(g):
            for _ in range(50): fn()
    torch.cuda.synchronize()
    return timeit(g.replay) / 50
for m, n_out in ((837, 600), (837, 300), (2903, 128), (5924, 128)):
    for k in (32, 128, 288, 608, 1216):
        x = torch.randn(m, k, device=dev); w = torch.randn(n_out, k, device=dev) / k ** 0.5; b = torch.randn(n_out, device=dev)
        st = torch.zeros(2 * n_out, dtype=torch.float64, device=dev)
        wt = w.t().contiguous().t()
        out = {"plain": graphed(lambda: layers._linear_hip([(x, None)], w, b, None, None, None, 1, m)),
               "stats": graphed(lambda: layers._linear_hip([(x, None)], w, b, None, None, None, 0, m, stats=st)),
               "transposed_w": graphed(lambda: layers._linear_hip([(x, None)], wt, b, None, None, None, 0, m))}
        print("M %5d N %4d K %5d: us per launch inside a replayed graph of 50: %s" % (m, n_out, k, {a: round(v, 2) for a, v in out.items()}), flush=True)

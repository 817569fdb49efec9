"""Embedding forward / backward at the config-4 edge shape (214 500 rows, bond tables 5 + 6 + 2 rows, d = 300) and the atom shape (105 083 rows, nine tables)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gsn_amd import encoding
dev = torch.device("cuda", 0)
def timeit(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, dims in ((214500, [5, 6, 2]), (105083, [119, 4, 12, 12, 10, 6, 6, 2, 2]), (214500, [3, 3, 3, 3])):
    torch.manual_seed(0)
    m = encoding.multi_embedding(list(dims), 300, "sum").to(dev)
    codes = torch.stack([torch.randint(0, n, (rows,)) for n in dims], 1).to(dev)
    with torch.no_grad():
        fwd = timeit(lambda: m(codes))
    g = torch.randn(rows, 300, device=dev)
    def fb():
        m.zero_grad(set_to_none=True)
        m(codes).backward(g)
    both = timeit(fb)
    byt = rows * 300 * 4
    print("rows %7d tables %-40s fwd %7.1f us (%.2f TB/s of output)  fwd+bwd %7.1f us" % (rows, dims, fwd, byt / fwd / 1e6, both), flush=True)

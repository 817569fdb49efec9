"""Does the training step read uninitialised memory?  Poison the caching allocator's free blocks with huge values / NaN, then compare two
eager runs and a graphed run of the same seed (tests/test_graphed_train_gpu.py's helpers)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import test_graphed_train_gpu as T

def poison(val):
    blocks = [torch.full((s,), val, device="cuda") for s in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12, 1 << 10) for _ in range(3)]
    del blocks
    torch.cuda.synchronize()

kind = sys.argv[1] if len(sys.argv) > 1 else "zinc"
for val in (None, 1e30, float("nan")):
    if val is not None:
        poison(val)
    sa, la = T._run(kind, "sgd", 4)
    if val is not None:
        poison(val)
    sa2, _ = T._run(kind, "sgd", 4)
    if val is not None:
        poison(val)
    sb, lb = T._run(kind, "sgd", 4, warm=2)
    bad = [k for k in sa if sa[k].is_floating_point() and not torch.isfinite(sa[k]).all()]
    print(kind, "poison", val, "eager-eager %.3g  eager-graph %.3g  non-finite tensors %d" % (T._rel(sa, sa2), T._rel(sa, sb), len(bad)), bad[:3], [float(x) for x in la])

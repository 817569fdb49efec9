"""Which Python lines issue the ATen ops (and which of them launch) in the full-model eval step of bench.py (count + GNNSubstructures forward)."""
import os, sys, collections, traceback
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from torch.utils._python_dispatch import TorchDispatchMode
import bench
dev = torch.device("cuda", 0)
step, G = bench.full_model_closure(dev, int(sys.argv[1]) if len(sys.argv) > 1 else 16384)
for _ in range(3): step()
torch.cuda.synchronize()
cnt = collections.Counter()
SKIP = ("aten.view", "aten.detach", "aten.t.", "aten.slice", "aten.select", "aten._unsafe_view", "aten.as_strided", "aten.unsqueeze",
        "aten.expand", "aten.alias", "aten.reshape", "aten.transpose", "aten.squeeze", "aten.permute", "aten.is_", "aten.sym_", "aten.lift_fresh", "aten._local_scalar")
class Tr(TorchDispatchMode):
    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            fr = [f for f in traceback.extract_stack() if "/gsn_amd/" in f.filename or f.filename.endswith("bench.py")]
            where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr[-3:])) if fr else "(torch)"
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), None)
            cnt[(name, where, shp)] += 1
        return func(*args, **(kwargs or {}))
with Tr():
    step()
torch.cuda.synchronize()
print("graphs", G, "ops:", sum(cnt.values()))
for (name, where, shp), c in sorted(cnt.items(), key=lambda kv: (kv[0][0].startswith("aten.empty"), -kv[1])):
    print("%4d  %-30s %-18s %s" % (c, name, shp, where[:200]))

"""Which Python lines issue the ATen glue ops of a training step (TorchDispatchMode + traceback; backward in the calling thread)."""
import os, sys, types, collections, traceback
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
which = sys.argv[1] if len(sys.argv) > 1 else "zinc"
import importlib
from torch.utils._python_dispatch import TorchDispatchMode
m = importlib.import_module("train_step_" + which)
dev = torch.device("cuda", 0)
args = types.SimpleNamespace(batch=128 if which == "zinc" else 32, layers=5, d=300, optimizer="sgd")
model, data, params, opt, loss_of, N, E = m.build(args, dev, 0)
def step():
    opt.zero_grad(set_to_none=True)
    loss = loss_of(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
cnt = collections.Counter()
SKIP = ("aten.view", "aten.empty", "aten.detach", "aten.t.", "aten.slice", "aten.select", "aten._unsafe_view", "aten.as_strided", "aten.unsqueeze",
        "aten.expand", "aten.alias", "aten.reshape", "aten.transpose", "aten.squeeze", "aten.permute", "aten.is_", "aten.sym_", "aten.lift_fresh", "aten._local_scalar")
class Tr(TorchDispatchMode):
    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            fr = [f for f in traceback.extract_stack() if "/gsn_amd/" in f.filename or "/scripts/train" in f.filename]
            where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr[-3:])) if fr else "(torch)"
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), None)
            cnt[(name, where, shp)] += 1
        return func(*args, **(kwargs or {}))
torch.autograd.set_multithreading_enabled(False)
with Tr():
    step()
torch.cuda.synchronize()
print(which, "N", N, "E", E, "ops:", sum(cnt.values()))
for (name, where, shp), c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%4d  %-28s %-18s %s" % (c, name, shp, where[:170]))

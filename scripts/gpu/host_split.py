#!/usr/bin/env python3
"""Host time per autograd Function of a small-batch training step (perf_counter around forward / backward of the custom Functions; no profiler)."""
import os, sys, time, types, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gsn_amd import layers, encoding
import importlib.util
spec = importlib.util.spec_from_file_location("tsm", os.path.join(ROOT, "scripts", "train_step_molhiv.py"))
tsm = importlib.util.module_from_spec(spec); spec.loader.exec_module(tsm)
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(cls, name):
    for meth in ("forward", "backward"):
        f = getattr(cls, meth)
        def g(*a, _f=f, _k=name + "." + meth, **k):
            t0 = time.perf_counter(); r = _f(*a, **k); dt = time.perf_counter() - t0
            acc[_k][0] += dt; acc[_k][1] += 1
            return r
        setattr(cls, meth, staticmethod(g))
for cls, name in ((layers._DenseStagesFn, "dense"), (layers._PropagateFn, "propagate"), (encoding._EmbedFn, "embed"), (layers._AddByGraphFn, "addbygraph")):
    wrap(cls, name)
args = types.SimpleNamespace(batch=int(sys.argv[1]) if len(sys.argv) > 1 else 32, steps=100, warmup=10, layers=5, d=300)
r = tsm.run(args, torch.device("cuda", 0))
n = args.steps + args.warmup
print("ms_per_step", r["ms_per_step"])
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-22s %6.3f ms/step  %5.1f calls/step  %6.1f us/call" % (k, t / n * 1e3, c / n, t / c * 1e6))

"""cProfile of the eager training step at the reference's batch size (host time per step by function)."""
import cProfile, pstats, io, os, sys, types
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
which = sys.argv[1] if len(sys.argv) > 1 else "molhiv"
import importlib
m = importlib.import_module("train_step_" + which)
dev = torch.device("cuda", 0)
args = types.SimpleNamespace(batch=128 if which == "zinc" else 32, layers=5, d=300, optimizer="sgd")
model, data, params, opt, loss_of, N, E = m.build(args, dev, 0)
torch.autograd.set_multithreading_enabled(False)       # backward in this thread: visible to the profiler
def step():
    opt.zero_grad(set_to_none=True)
    loss = loss_of(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(which, "enqueue ms/step %.3f, with sync %.3f" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(45)
out = s.getvalue().splitlines()
print("\n".join(l[:170] for l in out[:70]))

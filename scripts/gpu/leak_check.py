"""Long loops of the training step (eager and replayed) and of the headline forward: device / pinned memory must not grow."""
import os, sys, types, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import train_step_molhiv as tm
from gsn_amd import encoding, layers
from gsn_amd.graphs import GraphedTrainStep
dev = torch.device("cuda", 0)
model, data, params, opt, loss_of, N, E = tm.build(types.SimpleNamespace(batch=32, layers=5, d=300, optimizer="sgd"), dev, 0)
def step():
    opt.zero_grad(set_to_none=True)
    loss = loss_of(); loss.backward(); opt.step()
    return loss
for _ in range(50): step()
torch.cuda.synchronize()
m0, r0 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
meta0 = len(encoding._META_CACHE)
t0 = time.time()
for i in range(2000): step()
torch.cuda.synchronize()
print("eager 2000 steps %.1f s: allocated %+d B, reserved %+d B, meta cache %d -> %d, pinned ring words %d" % (
    time.time() - t0, torch.cuda.memory_allocated() - m0, torch.cuda.memory_reserved() - r0, meta0, len(encoding._META_CACHE), encoding._META_HOST[1]))
g = GraphedTrainStep(loss_of, opt, params, warmup=2)
torch.cuda.synchronize()
m1, r1 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
for i in range(20000): g()
torch.cuda.synchronize()
print("20000 replays: allocated %+d B, reserved %+d B, loss %.4f" % (torch.cuda.memory_allocated() - m1, torch.cuda.memory_reserved() - r1, float(g.loss)))
model.eval()
with torch.no_grad():
    for _ in range(20): model(data)
    torch.cuda.synchronize()
    m2 = torch.cuda.memory_allocated()
    for _ in range(3000): model(data)
    torch.cuda.synchronize()
print("3000 eval forwards: allocated %+d B" % (torch.cuda.memory_allocated() - m2))

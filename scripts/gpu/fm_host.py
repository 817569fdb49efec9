"""full_model_step of bench.py (count + 4-layer GNNSubstructures eval forward, 16 384 graphs): host enqueue time vs wall vs HIP events vs one replayed HIP graph."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from gsn_amd.graphs import GraphedStep
dev = torch.device("cuda", 0)
step, G = bench.full_model_closure(dev, int(os.environ.get("G", "16384")), check=False)
for _ in range(20): step()
torch.cuda.synchronize()
bench.spin_up(step)
t0 = time.perf_counter()
for _ in range(50): step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): step()
e1.record(); torch.cuda.synchronize()
print("eager: host enqueue %.4f ms/step, wall %.4f, HIP events %.4f" % (t_enq / 50 * 1e3, t_all / 50 * 1e3, e0.elapsed_time(e1) / 50))
g = GraphedStep(step, warmup=3, device=dev)
for _ in range(30): g()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): g()
torch.cuda.synchronize()
print("hip graph: %.4f ms/step" % ((time.perf_counter() - t0) / 50 * 1e3))

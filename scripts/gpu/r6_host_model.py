"""r06: where the host time of an EAGER forward of the config-2 model goes at the reference's batch size (B = 128): cProfile by function."""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "128"))
mstep, _ = bench.full_model_closure(dev, B, check=False)
for _ in range(20):
    mstep()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    mstep()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("B %d: enqueue %.1f us per forward, with sync %.1f" % (B, (t1 - t0) / 100 * 1e6, (t2 - t0) / 100 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    mstep()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print("\n".join(l[:160] for l in s.getvalue().splitlines()[:60]))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print("\n".join(l[:160] for l in s.getvalue().splitlines()[:65]))

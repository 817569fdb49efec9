#!/usr/bin/env python3
"""Every timed launch of one full-model step (bench.full_model_closure, 16 384 graphs): family, work (flops or bytes), ms, rate."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                      # noqa: E402
import bench                      # noqa: E402
from gsn_amd import flags, layers        # noqa: E402

dev = torch.device("cuda", 0)
step, G = bench.full_model_closure(dev, 16384)
for _ in range(5):
    step()
torch.cuda.synchronize()
flags.KERNEL_TIMER = {}
step()
torch.cuda.synchronize()
timer, flags.KERNEL_TIMER = flags.KERNEL_TIMER, None
rows = []
for k, evs in timer.items():
    for a, b, w in evs:
        ms = a.elapsed_time(b)
        rows.append((k, w or 0.0, ms))
for k, w, ms in rows:
    print("%-16s work %.3e  %.4f ms  %.1f T/s" % (k, w, ms, w / ms / 1e9 if ms > 0 else 0))

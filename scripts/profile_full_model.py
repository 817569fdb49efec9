#!/usr/bin/env python3
"""Per-kernel time of the supplementary full-model step of bench.py (count + 4-layer GNNSubstructures eval forward, 16 384 graphs):
which launches the K = 260 layers spend their time in.  Prints one JSON line (HIP-event family timers of gsn_amd.layers)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                      # noqa: E402
import bench                      # noqa: E402
from gsn_amd import flags, layers        # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    step, G = bench.full_model_closure(dev, 16384)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    flags.KERNEL_TIMER = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        step()
    e1.record()
    torch.cuda.synchronize()
    timer, flags.KERNEL_TIMER = flags.KERNEL_TIMER, None
    fam = {}
    for k, evs in timer.items():
        fam[k] = round(sum(a.elapsed_time(b) for a, b, _w in evs) / 10, 4)
    print(json.dumps({"graphs": G, "ms_per_step": round(e0.elapsed_time(e1) / 10, 4), "ms_by_family": fam}))


if __name__ == "__main__":
    main()

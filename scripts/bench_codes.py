#!/usr/bin/env python3
"""Layer-0 forward of BASELINE config 2 at 65536 graphs: dense one-hot inputs vs integer codes (layers.Codes).
Prints per-kernel-family times from HIP events on the launch stream."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from gsn_amd import flags, layers  # noqa: E402


def main():
    dev = "cuda"
    b = bench.make_batch(int(os.environ.get("G", "65536")), 5)
    N, E = b.num_nodes, b.num_edges
    ei = torch.from_numpy(b.edge_index).to(dev)
    xc = layers.Codes(torch.from_numpy(b.atom_type).to(dev), [28])
    ec = layers.Codes(torch.from_numpy(b.bond_type).to(dev), [4])
    ic = layers.Codes(torch.randint(0, 3, (E, 4), device=dev), [3, 3, 3, 3])
    torch.manual_seed(0)
    lay = layers.GSN_edge_sparse(**bench.CTOR).to(dev).eval()
    flags.CODE_STATUS_CHECK = False
    deg = torch.zeros(N, device=dev)
    out = []
    for name, a in (("dense", (xc.dense(), ic.dense(), ec.dense())), ("codes", (xc, ic, ec))):
        def run():
            with torch.no_grad():
                return lay(a[0], ei, identifiers=a[1], degrees=deg, edge_features=a[2])
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        flags.KERNEL_TIMER = {}
        reps = 10
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        kt = {k: round(sum(x.elapsed_time(y) for x, y, _ in v) / reps, 4) for k, v in flags.KERNEL_TIMER.items()}
        flags.KERNEL_TIMER = None
        out.append({"inputs": name, "N": N, "E": E, "kernel_ms": kt, "total_ms": round(sum(kt.values()), 4)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()

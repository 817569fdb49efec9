#!/usr/bin/env python3
"""HBM fraction of the stand-alone propagate kernel: out[t] = sum_{e->t} msg_e at 65536 ZINC-shaped
graphs.  Algorithmic bytes per launch = 8*E (source ids) + 4*E (perm) + 4*(N+1) (seg_ptr) + 4*[N*d_a + R_b*d_b + E*d_c] + 4*N*d_out."""
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
    out = []
    cases = [("gin x_j only d=128", 0, 128, 0, 0, False), ("gin cat(x_j, id_e, ef_e) 64+40+8", 0, 64, 40, 8, False),
             ("gin cat(x_j, id_j) global 64+40", 0, 64, 40, 0, True), ("ogb relu(x_j+id_e+ef_e) d=300", 1, 300, 300, 300, False),
             ("scatter-add of messages d=128 (b only)", 0, 0, 128, 0, False)]
    for name, kind, da, db, dc, per_node in cases:
        a = torch.randn(N, da, device=dev) if da else None
        bb = torch.randn(N if per_node else E, db, device=dev) if db else None
        c = torch.randn(E, dc, device=dev) if dc else None
        f = lambda: layers.propagate(kind, ei, 1, N, a=a, b=bb, c=c, b_per_node=per_node)
        with torch.no_grad():
            y = f()
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            flags.KERNEL_TIMER = {}
            for _ in range(10):
                f()
            torch.cuda.synchronize()
        evs = flags.KERNEL_TIMER.get("propagate_fwd", [])
        flags.KERNEL_TIMER = None
        ms = sum(x.elapsed_time(z) for x, z, _ in evs) / max(len(evs), 1)
        d_out = y.shape[1]
        byt = 12.0 * E + 4.0 * (N + 1) + 4.0 * (N * da + (N if per_node else E) * db + E * dc) + 4.0 * N * d_out
        out.append({"case": name, "N": N, "E": E, "ms": round(ms, 4), "GBps": round(byt / ms / 1e6, 1), "hbm_frac": round(byt / ms / 1e6 / 8000.0, 3)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A/B on one box: counting + identifier encoding as two launches (gsn_count_hip, gsn_one_hot_hip) against the fused
gsn_count_encode_hip, ZINC-shaped batch, cycle k = 3..6, GSN-e.  Prints one JSON line (ms per pass, HIP events)."""
import json
import os
import sys

import networkx as nx
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsn_amd import layers, synth                      # noqa: E402
from gsn_amd.counting import CountPlan, count_batch    # noqa: E402


def main():
    G = int(os.environ.get("GRAPHS", "65536"))
    dev = torch.device("cuda", 0)
    b = synth.zinc_shape_batch(G, seed=5)
    plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", False)
    node_ptr, edge_ptr, ei = (torch.from_numpy(a).to(dev) for a in (b.node_ptr, b.edge_ptr, b.edge_index))
    E = b.num_edges
    mn, me = int(max(b.node_ptr[1:] - b.node_ptr[:-1])), int(max(b.edge_ptr[1:] - b.edge_ptr[:-1]))
    ids = torch.empty((E, 4), dtype=torch.int64, device=dev)
    enc = torch.empty((E, 12), dtype=torch.float32, device=dev)
    kw = dict(ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, check=False)

    def two():
        count_batch(plan, node_ptr, edge_ptr, ei, out=ids, **kw)
        return layers.one_hot_identifiers(ids, [3, 3, 3, 3], clamp=True)

    def count_only():
        count_batch(plan, node_ptr, edge_ptr, ei, out=ids, **kw)

    def fused():
        count_batch(plan, node_ptr, edge_ptr, ei, encode=([3, 3, 3, 3], True), counts=False, encoded_out=enc, **kw)

    def fused_both():
        count_batch(plan, node_ptr, edge_ptr, ei, out=ids, encode=([3, 3, 3, 3], True), encoded_out=enc, **kw)

    res = {}
    for rep in range(2):
        for name, fn in (("count_only", count_only), ("count_then_one_hot", two), ("count_encode", fused), ("count_encode_and_int64", fused_both)):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name] = round(min(res.get(name, 1e9), e0.elapsed_time(e1) / 50), 4)
    assert torch.equal(enc, two())
    print(json.dumps({"graphs": G, "E": E, "ms": res}))


if __name__ == "__main__":
    main()

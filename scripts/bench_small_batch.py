#!/usr/bin/env python3
"""Real batch sizes of the reference (32 / 128 graphs per step): wall time per step of count + encode + layer-0 forward.
At these sizes the kernels take ~20 us and the step is bound by host-side enqueue work."""
import cProfile
import json
import os
import pstats
import sys
import time

import numpy as np
import torch
import networkx as nx

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from gsn_amd import layers  # noqa: E402
from gsn_amd.counting import CountPlan, count_batch  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", False)
    plan.device_table(dev)
    torch.manual_seed(0)
    layer = layers.GSN_edge_sparse(**bench.CTOR).to(dev).eval()
    out = []
    prof = os.environ.get("PROFILE")
    for G in (32, 128, 1024):
        b = bench.make_batch(G, seed=G)
        N, E = b.num_nodes, b.num_edges
        mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
        node_ptr, edge_ptr = torch.from_numpy(b.node_ptr).to(dev), torch.from_numpy(b.edge_ptr).to(dev)
        ei = torch.from_numpy(b.edge_index).to(dev)
        x = torch.nn.functional.one_hot(torch.from_numpy(b.atom_type), 28).float().to(dev)
        ef = torch.nn.functional.one_hot(torch.from_numpy(b.bond_type), 4).float().to(dev)
        deg = torch.zeros(N, device=dev)
        ids_out = torch.empty((E, plan.n_cols), dtype=torch.int64, device=dev)

        def step():
            layers._CSR_CACHE.clear()
            count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, out=ids_out, check=False)
            idf = layers.one_hot_identifiers(ids_out, [3, 3, 3, 3], clamp=True)
            with torch.no_grad():
                return layer(x, ei, identifiers=idf, degrees=deg, edge_features=ef)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        reps = 300
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        rec = {"graphs": G, "N": N, "E": E, "us_per_step": round(t_all / reps * 1e6, 1), "host_enqueue_us": round(t_enq / reps * 1e6, 1),
               "graphs_per_s": round(G * reps / t_all, 1)}
        # the same step captured once into a HIP graph (fixed shapes: e.g. serving, or the SR isomorphism test) and replayed
        try:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                y_static = step()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            t_g = time.perf_counter() - t0
            y_ref = step()
            torch.cuda.synchronize()
            rec["hip_graph_us_per_step"] = round(t_g / reps * 1e6, 1)
            rec["hip_graph_matches"] = bool(torch.equal(y_static, y_ref))
        except Exception as e:       # capture is best effort: report why it failed
            rec["hip_graph_error"] = str(e)[:200]
        out.append(rec)
        if prof and G == 128:
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(200):
                step()
            pr.disable()
            torch.cuda.synchronize()
            st = pstats.Stats(pr)
            st.sort_stats("tottime").print_stats(22)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

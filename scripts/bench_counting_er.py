#!/usr/bin/env python3
"""BASELINE configs[4]: counting-only throughput on synthetic Erdos-Renyi graphs G(n=128, m=1000) with the 21 connected
five-vertex patterns (all_simple_graphs k=5: 58 vertex-orbit / 56 edge-orbit columns).  Graphs are independent units:
every rank counts its own shard, no collective (weak scaling; barrier + max-over-ranks time only).

    python scripts/bench_counting_er.py [--gpus N] [--graphs 1024] [--mode vertex|edge]     (starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_counting_er.py --gpus N"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gsn_amd import synth, dist as gdist, patterns  # noqa: E402
from gsn_amd.counting import CountPlan, count_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=1024, help="graphs per launch per GPU")
    ap.add_argument("--mode", default="vertex")
    ap.add_argument("--induced", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--gpus", type=int, default=1)
    args = ap.parse_args()
    if args.gpus > 1 and not gdist.under_launcher():
        raise SystemExit(gdist.launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rank, world, local, dist = gdist.init_from_env("nccl", dev)
    if args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "orbits.npz"))
    pats = [z["all_simple_graphs_5/%d/edges" % i].tolist() for i in range(21)]
    plan = CountPlan.get(pats, args.mode, bool(args.induced))
    graphs = [synth.er_graph(128, 1000, 100000 * rank + s) for s in range(args.graphs)]      # seeds as SURVEY 8(d)-5
    b = synth.collate(graphs)
    node_ptr, edge_ptr = torch.from_numpy(b.node_ptr).to(dev), torch.from_numpy(b.edge_ptr).to(dev)
    ei = torch.from_numpy(b.edge_index).to(dev)
    rows = b.num_edges if args.mode == "edge" else b.num_nodes
    out = torch.empty((rows, plan.n_cols), dtype=torch.int64, device=dev)
    # graphs handed to the kernel by falling estimated cost (sum_v deg^(k-1)): the long searches start first (scripts/gpu/r6_er_order.py);
    # GSN_ER_ORDER=0: as they come
    order = None
    if os.environ.get("GSN_ER_ORDER", "1") != "0":
        cost = np.asarray(gdist.counting_cost(b.edge_index, b.edge_ptr, 5), dtype=np.float64)
        order = torch.from_numpy(np.argsort(-cost, kind="stable").astype(np.int32)).to(dev)
    f = lambda: count_batch(plan, node_ptr, edge_ptr, ei, ids_are_global=True, max_nodes=128, max_edges=int(np.diff(b.edge_ptr).max()),
                            device=dev, out=out, check=False, graph_ids=order)
    f()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = gdist.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        # work figures from the output alone: occurrence-positions = sum of counts; maps = sum_p aut_p *
        # (pattern p's column sums) / k_p (vertex mode) or / (2 |E(H_p)|) (edge mode)
        cs = out.sum(dim=0).cpu().numpy().astype(np.float64)
        maps, c0 = 0.0, 0
        for el in pats:
            info = patterns.analyse(el, False)
            w = info["n_edge_orbits"] if args.mode == "edge" else info["n_vertex_orbits"]
            maps += info["aut_count"] * cs[c0:c0 + w].sum() / float(len(info["arcs"]) if args.mode == "edge" else info["k"])
            c0 += w
        print(json.dumps({"workload": "ER G(128,1000) x%d graphs/GPU, 21 five-vertex patterns, %s mode, induced=%d" % (args.graphs, args.mode, args.induced),
                          "n_gpus": world, "graphs_per_s": round(world * args.graphs * args.steps / dt, 1),
                          "ms_per_launch": round(dt / args.steps * 1e3, 3), "columns": plan.n_cols,
                          "occurrence_positions_per_s": round(world * float(cs.sum()) * args.steps / dt, 1),
                          "maps_per_s": round(world * maps * args.steps / dt, 1), "scaling": "weak",
                          "note": "rank 0's work figures x n_gpus (every rank counts a different, equally distributed shard)"}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

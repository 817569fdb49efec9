#!/usr/bin/env python3
"""BASELINE configs[1] as a TRAINING step (README.md:112 flags): GNNSubstructures, GSN_edge_sparse general, 4 layers, d=128,
one-hot atom / bond / identifier encoders, cycle counts k<=6 (GSN-e, local), sum readout, L1 loss, batch 128 per GPU; graph-shard
data parallel with one flat RCCL gradient all-reduce per step.  Forward and backward run on the HIP kernels.

    python scripts/train_step_zinc.py [--batch 128] [--steps 20]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/train_step_zinc.py"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch
import networkx as nx

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gsn_amd import dist as gdist, encoding, models, synth  # noqa: E402
from gsn_amd.counting import CountPlan, count_batch  # noqa: E402


def build(args, dev, rank=0):
    """(model, data, params, optimizer, loss closure, N, E) of the step."""
    b = synth.zinc_shape_batch(args.batch, seed=200 + rank)
    N, E = b.num_nodes, b.num_edges
    plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", False)
    ids, _ = count_batch(plan, b.node_ptr, b.edge_ptr, torch.from_numpy(b.edge_index), ids_are_global=True, device=dev)
    codes, d_id = encoding.unique_codes(ids)
    rng = np.random.default_rng(rank)
    data = types.SimpleNamespace(x=torch.from_numpy(b.atom_type).unsqueeze(1).to(dev), edge_index=torch.from_numpy(b.edge_index).to(dev),
                                 edge_features=torch.from_numpy(b.bond_type).unsqueeze(1).to(dev), identifiers=codes.to(dev),
                                 batch=torch.from_numpy(np.asarray(b.batch).astype(np.int64)).to(dev), degrees=torch.zeros(N, device=dev),
                                 y=torch.from_numpy(rng.standard_normal((args.batch, 1)).astype(np.float32)).to(dev))
    if os.environ.get("GSN_TRAIN_PARTITION", "1") != "0":
        # the collated batch's graph boundaries (what the counting kernel takes as well): the layers' aggregation index is then ONE launch per
        # direction (gsn_csr_build_graphs_hip) instead of the generic build, the readout needs none (models._register_partition)
        data.graph_partition = (torch.from_numpy(b.node_ptr.astype(np.int64)).to(dev), torch.from_numpy(b.edge_ptr.astype(np.int64)).to(dev),
                                int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max()), False)
    L, d = 4, 128
    act = getattr(args, "activation", "relu")      # (the tests of replay == eager use elu: no ReLU kink for last-bit noise to flip)
    kw = dict(seed=0, model_name="GSN_edge_sparse", readout="sum", dropout_features=[0.0] * (L + 1), bn=[True] * L,
              final_projection=[False] * L + [True], inject_ids=False, inject_edge_features=True, random_features=False,
              id_scope="local", d_msg=[d] * L, d_out=[d] * L, d_h=[[d]] * L, aggr="add", flow="source_to_target",
              msg_kind="general", train_eps=[False] * L, activation_mlp=act, bn_mlp=True, jk_mlp=True, degree_embedding="None",
              degree_as_tag=[False] * L, retain_features=[True] * L, multi_embedding_aggr="sum", input_node_encoder="one_hot_encoder",
              d_out_node_encoder=d, edge_encoder="one_hot_encoder", d_out_edge_encoder=[d] * L, id_embedding="one_hot_encoder",
              d_out_id_embedding=d, d_out_degree_embedding=d, extend_dims=True, activation=act)
    torch.manual_seed(0)
    model = models.GNNSubstructures(1, 1, None, d_id, 1, [28], [4], None, None, **kw).to(dev).train()
    params = list(model.parameters())
    if getattr(args, "optimizer", "sgd") == "adam":
        opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
    else:
        opt = torch.optim.SGD(params, lr=1e-3)
    loss_fn = torch.nn.L1Loss()
    return model, data, params, opt, (lambda: loss_fn(model(data), data.y)), N, E


def run(args, dev, dist=None, rank=0, world=1):
    """The timed steps; the result line as a dict (bench.py `train_small_batch` calls this)."""
    model, data, params, opt, loss_of, N, E = build(args, dev, rank)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = loss_of()
        loss.backward()
        gdist.allreduce_gradients(params, average=True)
        opt.step()
        return loss

    if getattr(args, "graph", False):
        from gsn_amd.graphs import GraphedTrainStep
        step = GraphedTrainStep(loss_of, opt, params, warmup=max(1, args.warmup))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_params = sum(p.numel() for p in params)
    return ({"workload": "ZINC-shaped GNNSubstructures training step (4 x GSN_edge_sparse general d=128), batch %d graphs/GPU (N=%d, E=%d)" % (args.batch, N, E),
            "n_gpus": world, "graphs_per_s": round(world * args.batch * args.steps / dt, 1),
            "ms_per_step": round(dt / args.steps * 1e3, 3), "parameters": n_params, "loss": float(loss.item()),
            "launch": "hip_graph" if getattr(args, "graph", False) else "eager", "optimizer": getattr(args, "optimizer", "sgd")})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graph", action="store_true", help="replay the whole step as one HIP graph (gsn_amd.graphs.GraphedTrainStep)")
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "adam"])
    args = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if "RANK" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    res = run(args, dev, dist, rank, world)
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

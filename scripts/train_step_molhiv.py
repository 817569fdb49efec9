#!/usr/bin/env python3
"""BASELINE configs[3] shape: ogbg-molhiv-like training step of the virtual-node model (gsn_amd.models.GNN_OGB,
README.md:121 flags: 5 layers, d=300, d_h=600, GSN_edge_sparse_ogb / msg_kind ogb, cycle ids k<=6 local, embedding
encoders, vn, batch 32 per GPU) with graph-shard data parallelism: every rank steps on its own batch, ONE flat-bucket RCCL
all-reduce of the gradients per step (gsn_amd.dist.allreduce_gradients), SGD update.

    python scripts/train_step_molhiv.py [--batch 32] [--steps 20]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/train_step_molhiv.py

Synthetic molhiv-shaped molecules (n ~ N(25.5, 12) clipped to [2, 222] is approximated by the ZINC-shape generator with
that mean / spread), random integer atom / bond features within ogb's feature dims, identifiers = real cycle counts from
the HIP counting kernel recoded with one_hot_unique.  Prints one JSON line from rank 0."""
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


def make_data(n_graphs, seed, dev):
    b = synth.zinc_shape_batch(n_graphs, seed=seed, mean_n=25.5, sd_n=8.0, n_min=4, n_max=60)
    rng = np.random.default_rng(seed)
    N, E = b.num_nodes, b.num_edges
    plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", True)
    ids, _ = count_batch(plan, b.node_ptr, b.edge_ptr, torch.from_numpy(b.edge_index), ids_are_global=True, device=dev)
    codes, d_id = encoding.unique_codes(ids)
    d = types.SimpleNamespace(
        x=torch.from_numpy(rng.integers(0, encoding.ATOM_FEATURE_DIMS, size=(N, 9))).to(dev),
        edge_index=torch.from_numpy(b.edge_index).to(dev),
        edge_features=torch.from_numpy(rng.integers(0, encoding.BOND_FEATURE_DIMS, size=(E, 3))).to(dev),
        identifiers=codes.to(dev), batch=torch.from_numpy(np.asarray(b.batch).astype(np.int64)).to(dev),
        degrees=torch.zeros(N, device=dev), y=torch.from_numpy(rng.integers(0, 2, size=(n_graphs, 1)).astype(np.float32)).to(dev))
    if os.environ.get("GSN_TRAIN_PARTITION", "1") != "0":
        # the collated batch's graph boundaries (what the counting kernel takes as well): the layers' aggregation index is then ONE launch per
        # direction (gsn_csr_build_graphs_hip) instead of the generic build, the readout needs none (models._register_partition)
        d.graph_partition = (torch.from_numpy(b.node_ptr.astype(np.int64)).to(dev), torch.from_numpy(b.edge_ptr.astype(np.int64)).to(dev),
                             int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max()), False)
    return d, d_id, N, E


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--layers", type=int, default=5)
    ap.add_argument("--d", type=int, default=300)
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


def build(args, dev, rank=0, dropout=0.5):
    """(model, data, params, optimizer, loss closure, N, E) of the step."""
    data, d_id, N, E = make_data(args.batch, 100 + rank, dev)
    L, dm = args.layers, args.d
    act = getattr(args, "activation", "relu")      # (the tests of replay == eager use elu in the MLPs; the ogb message keeps its relu)
    kw = dict(seed=0, model_name="GSN_edge_sparse_ogb", readout="mean", dropout_features=[dropout] * (L + 1), bn=[True] * L,
              final_projection=[False] * L + [True], residual=False, inject_ids=True, vn=True, id_scope="local",
              d_msg=[dm] * L, d_out=[dm] * L, d_h=[[2 * dm]] * L, aggr="add", flow="source_to_target", msg_kind="ogb",
              train_eps=[True] * L, activation_mlp=act, bn_mlp=True, jk_mlp=False, degree_embedding="None",
              degree_as_tag=[False] * L, retain_features=[True] * L, multi_embedding_aggr="sum", features_scope="full",
              input_node_encoder="atom_encoder", d_out_node_encoder=dm, input_vn_encoder="embedding", d_out_vn_encoder=dm,
              edge_encoder="bond_encoder", d_out_edge_encoder=[dm] * L, id_embedding="embedding", d_out_id_embedding=dm,
              d_out_degree_embedding=dm, d_out_vn=[dm] * (L - 1), vn_pooling="sum", extend_dims=True, activation=act)
    torch.manual_seed(0)      # identical replicas
    model = models.GNN_OGB(9, 1, None, d_id, 3, None, None, None, None, **kw).to(dev).train()
    params = [p for p in model.parameters()]
    n_params = sum(p.numel() for p in params)
    if getattr(args, "optimizer", "sgd") == "adam":
        opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
    else:
        opt = torch.optim.SGD(params, lr=1e-3)
    loss_fn = torch.nn.BCEWithLogitsLoss()
    return model, data, params, opt, (lambda: loss_fn(model(data), data.y)), N, E


def run(args, dev, dist=None, rank=0, world=1):
    """The timed training steps; returns the result line as a dict (bench.py's supplementary `train_step_config4` calls this with
    ``types.SimpleNamespace(batch=4096, steps=5, warmup=2, layers=5, d=300)``)."""
    model, data, params, opt, loss_of, N, E = build(args, dev, rank)
    L, dm = args.layers, args.d
    act = getattr(args, "activation", "relu")      # (the tests of replay == eager use elu in the MLPs; the ogb message keeps its relu)
    n_params = sum(p.numel() for p in params)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = loss_of()
        loss.backward()
        gdist.allreduce_gradients(params, average=True)      # one flat fp32 bucket, one RCCL all-reduce
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
    return {"workload": "molhiv-shaped GNN_OGB training step (%d layers, d=%d, vn), batch %d graphs/GPU (N=%d, E=%d)" % (L, dm, args.batch, N, E),
            "n_gpus": world, "graphs_per_s": round(world * args.batch * args.steps / dt, 1),
            "ms_per_step": round(dt / args.steps * 1e3, 3), "parameters": n_params,
            "grad_bucket_MB": round(n_params * 4 / 1e6, 2), "loss": float(loss.item()),
            "launch": "hip_graph" if getattr(args, "graph", False) else "eager", "optimizer": getattr(args, "optimizer", "sgd"),
            "note": "forward and backward on HIP kernels (native adjoints of every stage, DESIGN.md 4); SGD update and glue in PyTorch"}


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The RCCL path of the training step, checkable on any number of ranks INCLUDING ONE (a `gpurun` box has one GPU):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/dist_selfcheck.py

Every rank builds the same ZINC-shaped model (same seed) on its own shard, runs one forward + backward and then
gsn_amd.dist.allreduce_gradients(force=True) -- at world size 1 the collective still goes through RCCL.  Checks, per rank:
  * gradients after the all-reduce == mean over ranks of the gradients before it (all-gathered and reduced in fp64 on every rank;
    at world size 1: bit-for-bit the gradients before it),
  * parameters without a gradient on any rank keep grad = None,
  * the same step captured by GraphedTrainStep (force_allreduce=True: the RCCL all-reduce is a node of the HIP graph) replays and
    keeps the replicas identical.
Rank 0 prints one JSON line: RCCL version, bucket size, all-reduce time eager / inside the replayed step."""
import json
import os
import sys
import time
import types

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gsn_amd import dist as gdist  # noqa: E402
from gsn_amd.graphs import GraphedTrainStep  # noqa: E402
import train_step_zinc as tz  # noqa: E402


def main():
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    model, data, params, opt, loss_of, N, E = tz.build(types.SimpleNamespace(batch=64, optimizer="sgd"), dev, rank)
    opt.zero_grad(set_to_none=True)
    loss_of().backward()
    before = [None if p.grad is None else p.grad.detach().clone() for p in params]
    gdist.allreduce_gradients(params, average=True, force=True)
    torch.cuda.synchronize()
    ok_grad, ok_none = True, True
    for p, g0 in zip(params, before):
        have = torch.tensor([0.0 if g0 is None else 1.0], device=dev)
        dist.all_reduce(have)
        if float(have.item()) == 0:
            ok_none = ok_none and p.grad is None
            continue
        mine = (torch.zeros_like(p) if g0 is None else g0).double()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        want = torch.stack(gathered).sum(0) / world
        if world == 1:
            ok_grad = ok_grad and torch.equal(p.grad, g0)
        else:
            ok_grad = ok_grad and bool(((p.grad.double() - want).abs() <= 1e-6 * want.abs().max().clamp_min(1e-30)).all())
    # eager all-reduce time of this bucket
    for _ in range(3):
        gdist.allreduce_gradients(params, average=True, force=True)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        gdist.allreduce_gradients(params, average=True, force=True)
    torch.cuda.synchronize()
    t_ar = (time.perf_counter() - t0) / 20
    # the captured step with the collective inside the graph
    note = None
    try:
        step = GraphedTrainStep(loss_of, opt, params, warmup=2, force_allreduce=True)
        for _ in range(3):
            loss = step()
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            loss = step()
        torch.cuda.synchronize()
        t_step = (time.perf_counter() - t0) / 20
        # replicas stay identical: checksum of the parameters, max - min over ranks
        cs = torch.stack([p.detach().double().sum() for p in params]).sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ok_replica = bool((hi - lo).abs().item() <= 1e-9 * abs(float(hi.item())) + 1e-12)
        finite = bool(torch.isfinite(loss).item())
    except Exception as ex:      # reported, not hidden: the eager path above is the contract
        t_step, ok_replica, finite, note = None, False, False, "capture with RCCL inside failed: " + str(ex)[:300]
    if rank == 0:
        n_par = sum(p.numel() for p in params if p.requires_grad)
        print(json.dumps({"world": world, "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()), "bucket_MB": round(n_par * 4 / 1e6, 3),
                          "grads_equal_mean_over_ranks": ok_grad, "untouched_parameters_keep_none": ok_none,
                          "allreduce_gradients_ms": round(t_ar * 1e3, 4), "graphed_step_with_rccl_ms": None if t_step is None else round(t_step * 1e3, 4),
                          "replicas_identical_after_replays": ok_replica, "loss_finite": finite, "note": note}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

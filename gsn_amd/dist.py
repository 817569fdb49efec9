"""Multi-GPU plumbing for the two hot paths: one process per GPU, graphs sharded across ranks.

Counting and forward message passing need no exchange (a batch is a disjoint union of graphs); training needs exactly
one gradient all-reduce per optimizer step.  Gradient volume is tiny (<= ~13 MB for the molhiv model), so the
step is latency-bound over xGMI: one flat fp32 bucket, one all-reduce (RCCL via backend "nccl"; "gloo" on CPU tests).

Launching: ``launch_ranks`` re-executes a script under ``python -m torch.distributed.run`` (one rank per GPU of this node,
rendezvous on 127.0.0.1), so ``python bench.py --gpus N`` and ``torchrun ... bench.py --gpus N`` run the same code.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist


# ----------------------------------------------------------------------------------------------------------------------
# process launch / rendezvous
# ----------------------------------------------------------------------------------------------------------------------
def under_launcher() -> bool:
    """True inside a process started by torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK in the environment)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n_ranks: int, script: str, argv, env=None) -> int:
    """Run ``script argv`` as ``n_ranks`` processes of ONE node under torch.distributed.run and return its exit code;
    the ranks' stdout / stderr pass through (rank 0 prints the result line)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_ranks)),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL needs it)
    e.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=e)


def init_from_env(backend: str = "nccl", device=None):
    """(rank, world, local_rank, dist-or-None) from the launcher's environment; initialises the process group when the
    process was started by torch.distributed.run (also with one rank).  ``backend`` "nccl" = RCCL on ROCm."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not under_launcher():
        return rank, world, local_rank, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank, dist


def max_over_ranks(value: float, device=None) -> float:
    """MAX of a host scalar over all ranks (the step time every rank reports is the slowest rank's)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ----------------------------------------------------------------------------------------------------------------------
# sharding
# ----------------------------------------------------------------------------------------------------------------------
def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) slice of ``n_items`` for ``rank`` (first ``n_items % world`` ranks get one more)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def counting_cost(edge_index, edge_ptr, k: int):
    """Per-graph cost proxy of the counting kernel, SURVEY.md 8(e): sum_v deg(v)^(k-1) -- the number of k-vertex walks a
    rooted search can open from v.  edge_index int64 [2, E_total] (both directions present; any vertex numbering in which
    graphs do not share ids), edge_ptr int64 [G+1] -> float64 [G]."""
    ei = np.asarray(edge_index)
    ept = np.asarray(edge_ptr, dtype=np.int64)
    G = len(ept) - 1
    if G <= 0:
        return np.zeros(0, dtype=np.float64)
    if ei.shape[1] == 0:
        return np.ones(G, dtype=np.float64)
    _, inv, deg = np.unique(ei[0], return_inverse=True, return_counts=True)
    # sum_v deg(v)^(k-1) = sum over columns (u, .) of deg(u)^(k-2): every vertex owns deg(v) columns
    per_col = deg.astype(np.float64)[inv] ** max(int(k) - 2, 0)
    csum = np.concatenate([[0.0], np.cumsum(per_col)])
    return csum[ept[1:]] - csum[ept[:-1]] + 1.0


def shard_by_cost(costs, world: int):
    """Split items (in order) into ``world`` contiguous chunks of roughly equal total cost; returns world+1 boundaries.
    Use with :func:`counting_cost`."""
    c = torch.as_tensor(np.asarray(costs, dtype=np.float64)).cumsum(0)
    total = float(c[-1]) if len(c) else 0.0
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(torch.searchsorted(c, torch.tensor(total * r / world, dtype=torch.float64))))
    bounds.append(len(c))
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


# ----------------------------------------------------------------------------------------------------------------------
# training: one flat-bucket gradient all-reduce per optimizer step
# ----------------------------------------------------------------------------------------------------------------------
_BUCKETS = {}


def allreduce_gradients(parameters, average: bool = True, group=None):
    """Sum (or average) the gradients of ``parameters`` over all ranks with ONE all-reduce of a flat bucket.

    The bucket covers EVERY parameter that requires grad -- a parameter whose ``.grad`` is None on this rank (unused
    branch, empty shard) contributes zeros and receives the reduced value -- so all ranks issue an identical collective
    whatever their local graphs exercised.  The bucket is allocated once per parameter list and reused."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return
    dtype, device = params[0].dtype, params[0].device
    for p in params:
        if p.dtype != dtype or p.device != device:
            raise TypeError("allreduce_gradients: parameters must share one dtype and device (got %s/%s and %s/%s)"
                            % (dtype, device, p.dtype, p.device))
    key = (tuple(id(p) for p in params), dtype, device)
    # behind the gradients: one has-gradient flag per parameter (summed by the same collective), so that a parameter no rank produced a
    # gradient for keeps ``grad = None`` -- with zeros instead, weight decay / momentum would move parameters a single-GPU run never touches
    n_grad = sum(p.numel() for p in params)
    n_total = n_grad + len(params)
    flat = _BUCKETS.get(key)
    if flat is None or flat.numel() != n_total:
        if len(_BUCKETS) > 8:
            _BUCKETS.clear()
        flat = torch.empty(n_total, dtype=dtype, device=device)
        _BUCKETS[key] = flat
    off = 0
    for i, p in enumerate(params):
        n = p.numel()
        if p.grad is None:
            flat[off:off + n].zero_()
        else:
            flat[off:off + n].copy_(p.grad.reshape(-1))
        flat[n_grad + i] = 0.0 if p.grad is None else 1.0
        off += n
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    has = flat[n_grad:].tolist()                      # (one small read-back per step; the gradients themselves stay on the device)
    if average:
        flat[:n_grad] /= dist.get_world_size(group)
    off = 0
    for i, p in enumerate(params):
        n = p.numel()
        if has[i] == 0:
            pass                                      # no rank has a gradient: stays None, as on one GPU
        elif p.grad is None:
            p.grad = flat[off:off + n].view_as(p).clone()
        else:
            p.grad.copy_(flat[off:off + n].view_as(p))
        off += n

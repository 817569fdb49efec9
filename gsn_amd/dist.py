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


def require_devices(n: int) -> None:
    """Fail loudly, before any collective, when this node shows fewer GPUs than the ranks that were started on it."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        raise SystemExit("gsn_amd.dist: %d rank(s) were started on this node but torch.cuda.device_count() = %d "
                         "(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = %r / %r)"
                         % (n, have, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")))


def rank_report(device, group=None):
    """What makes a first multi-GPU run diagnosable: per rank its device ordinal, PCI bus id, device name and memory, the world size the
    process group reports, the backend -- gathered to every rank (list of dicts, rank order) with one all_gather_object."""
    info = {"rank": dist.get_rank(group) if dist.is_initialized() else 0, "pid": os.getpid(), "host": os.uname().nodename}
    if device is not None and torch.device(device).type == "cuda":
        idx = torch.device(device).index
        idx = torch.cuda.current_device() if idx is None else idx
        pr = torch.cuda.get_device_properties(idx)
        info.update(device=idx, current_device=torch.cuda.current_device(), name=pr.name, total_memory_GiB=round(pr.total_memory / 2 ** 30, 1),
                    multi_processor_count=pr.multi_processor_count, pci_bus_id=getattr(pr, "pci_bus_id", None),
                    pci_device_id=getattr(pr, "pci_device_id", None), gcn_arch=getattr(pr, "gcnArchName", None),
                    visible=os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES"))
    if not dist.is_available() or not dist.is_initialized():
        return [info]
    info.update(world_size=dist.get_world_size(group), backend=dist.get_backend(group))
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, info, group=group)
    return out


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


FLAG_STABLE_CALLS = 3        # (steady mode) eager calls with identical reduced has-gradient flags before the read-back becomes asynchronous
STEADY_FLAGS = os.environ.get("GSN_ALLREDUCE_STEADY", "0") != "0"      # default of allreduce_gradients(steady_flags=None)


class _Bucket:
    __slots__ = ("flat", "views", "n_grad", "has", "pinned", "calls", "stable", "late", "late_event")

    def __init__(self, params, dtype, device):
        self.n_grad = sum(p.numel() for p in params)
        self.flat = torch.zeros(self.n_grad + len(params), dtype=dtype, device=device)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.has = None          # which parameters some rank had a gradient for, as of the last eager call that exchanged the flags
        self.pinned = False      # a captured graph holds this bucket's addresses: never evicted
        self.calls = 0           # eager calls so far (identical on every rank: the schedule of flag exchanges is derived from it)
        self.stable = 0          # consecutive flag exchanges that returned the same flags
        self.late = None         # (steady mode) pinned host copy of the reduced flags of the last call, read at the NEXT call
        self.late_event = None


def allreduce_gradients(parameters, average: bool = True, group=None, force: bool = False, steady_flags=None):
    """Sum (or average) the gradients of ``parameters`` over all ranks with ONE all-reduce of a flat bucket.

    The bucket covers EVERY parameter that requires grad -- a parameter whose ``.grad`` is None on this rank (unused
    branch, empty shard) contributes zeros and receives the reduced value -- so all ranks issue an identical collective
    whatever their local graphs exercised.  The bucket is allocated once per parameter list and reused; gradients enter and
    leave it with one multi-tensor copy each way (not one launch per parameter).

    Behind the gradients ride one has-gradient flag per parameter, summed by the same collective IN EVERY CALL, so that a parameter NO rank
    produced a gradient for keeps ``grad = None`` -- with zeros instead, weight decay / momentum would move parameters a single-GPU
    run never touches.  By default the reduced flags are read back in every call (one small device-to-host read: exact whatever the
    batches exercise -- a data-dependent branch, a parameter first used after an epoch switch).  ``steady_flags=True``
    (``GSN_ALLREDUCE_STEADY=1``): after FLAG_STABLE_CALLS identical exchanges the read-back becomes asynchronous -- the reduced flags land in
    pinned memory behind an event and are compared at the NEXT call, the step itself uses the last known flags and does not synchronise.  A
    rank that holds a gradient the known flags do not list takes the synchronous path for that call by itself (the collective is the same
    either way); a rank that learns one call late that the set had grown raises RuntimeError (it kept ``grad = None`` where its peers
    stepped: the replicas are one update apart) -- never silently.  (r05 skipped the exchange itself for up to 63 calls: ADVICE r05.)
    Inside a stream capture (gsn_amd.graphs.GraphedTrainStep) nothing may be read back: the captured step reduces the gradient part only and
    reuses the flags of the last eager call with this parameter list (the warm-up steps in front of the capture), which is exact as long as
    the set of parameters that receive gradients is a property of the model, not of the batch.

    ``force``: run the collective at world size 1 too (tests: the RCCL path on one GPU)."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
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
    b = _BUCKETS.get(key)
    if b is None or b.flat.numel() != sum(p.numel() for p in params) + len(params):
        if len(_BUCKETS) > 8:
            # (a bucket that a captured step copies into and all-reduces lives at addresses the graph recorded: it stays)
            for k in [k for k, v in _BUCKETS.items() if not v.pinned]:
                del _BUCKETS[k]
        b = _BUCKETS[key] = _Bucket(params, dtype, device)
    capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    if capturing:
        b.pinned = True
    have = [p.grad is not None for p in params]
    present = [i for i, h in enumerate(have) if h]
    absent = [i for i, h in enumerate(have) if not h]
    with torch.no_grad():
        if present:
            torch._foreach_copy_([b.views[i] for i in present], [params[i].grad for i in present])
        if absent:
            torch._foreach_zero_([b.views[i] for i in absent])
        if capturing:
            if b.has is None:
                raise RuntimeError("allreduce_gradients inside a stream capture needs one eager call with the same parameters first "
                                   "(GraphedTrainStep's warm-up steps do that)")
            dist.all_reduce(b.flat[:b.n_grad], op=dist.ReduceOp.SUM, group=group)
            has = b.has
        else:
            b.calls += 1
            steady = STEADY_FLAGS if steady_flags is None else bool(steady_flags)
            if b.late_event is not None:
                # the reduced flags of the call before this one (asynchronous read-back): they must be the flags that call used
                b.late_event.synchronize()
                late, b.late_event = [v != 0 for v in b.late.tolist()], None
                if late != b.has:
                    grown = [i for i, (x, y) in enumerate(zip(late, b.has)) if x and not y]
                    b.has, b.stable = late, 0
                    if grown:
                        raise RuntimeError("allreduce_gradients(steady_flags=True): in the previous call parameters %r received a gradient on some rank for the "
                                           "first time; this rank kept grad = None for those it had none for while its peers stepped -- the replicas are one "
                                           "update apart.  Which parameters receive gradients depends on the batch here: call with steady_flags=False." % (grown,))
            b.flat[b.n_grad:].copy_(torch.tensor([1.0 if h else 0.0 for h in have], dtype=dtype), non_blocking=True)
            dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=group)
            locally_new = b.has is not None and any(h and not k for h, k in zip(have, b.has))
            if steady and device.type == "cuda" and b.has is not None and b.stable >= FLAG_STABLE_CALLS and not locally_new:
                if b.late is None:
                    b.late = torch.empty(len(params), dtype=dtype).pin_memory()
                b.late.copy_(b.flat[b.n_grad:], non_blocking=True)
                b.late_event = torch.cuda.Event()
                b.late_event.record()
                has = b.has
            else:
                has = [v != 0 for v in b.flat[b.n_grad:].tolist()]          # (one small read-back; the gradients stay on the device)
                b.stable = b.stable + 1 if has == b.has else 1
                b.has = has
        world = dist.get_world_size(group)
        if average and world > 1:
            b.flat[:b.n_grad] /= world
        if present:
            torch._foreach_copy_([params[i].grad for i in present], [b.views[i] for i in present])
        for i in absent:
            if has[i]:
                params[i].grad = b.views[i].clone()
            # else: no rank has a gradient: stays None, as on one GPU

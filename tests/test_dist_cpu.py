"""N>1 path on CPU: world_size-2 gloo.  Shards are disjoint and cover everything; the flat-bucket gradient all-reduce
equals the mean of the per-rank gradients."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gsn_amd.dist import shard_range, allreduce_gradients
    lo, hi = shard_range(1001, rank, world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    x = torch.arange(1001 * 5, dtype=torch.float32).reshape(1001, 5) / 1000.0
    model(x[lo:hi]).pow(2).sum().backward()           # every rank: its own shard of "graphs"
    local = [p.grad.clone() for p in model.parameters()]
    allreduce_gradients(model.parameters(), average=True)
    q.put((rank, lo, hi, [g.numpy() for g in local], [p.grad.numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_grad_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, l0, a0), (r1, lo1, hi1, l1, a1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 501, 501, 1001)
    for g0, g1, s0, s1 in zip(l0, l1, a0, a1):
        assert abs(s0 - s1).max() == 0
        assert abs(s0 - (g0 + g1) / 2).max() < 1e-5 * max(1.0, abs(s0).max())


def test_shard_by_cost():
    from gsn_amd.dist import shard_by_cost, shard_range
    b = shard_by_cost([1, 1, 1, 1, 100, 1, 1, 1], 2)
    assert b[0] == 0 and b[-1] == 8 and b == sorted(b)
    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]

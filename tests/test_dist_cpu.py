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
    extra = torch.nn.Parameter(torch.ones(4))         # used by rank 0 only: rank 1's .grad stays None
    unused = torch.nn.Parameter(torch.ones(3))        # used by NO rank: must keep grad = None (as on one GPU), not receive zeros
    loss = model(x[lo:hi]).pow(2).sum()
    if rank == 0:
        loss = loss + (extra * torch.arange(4.0)).sum()
    loss.backward()                                   # every rank: its own shard of "graphs"
    local = [p.grad.clone() for p in model.parameters()]
    allreduce_gradients(list(model.parameters()) + [extra, unused], average=True)
    assert extra.grad is not None and torch.allclose(extra.grad, torch.arange(4.0) / 2)   # zeros from the rank that had none
    assert unused.grad is None
    allreduce_gradients(list(model.parameters()) + [extra, unused], average=False)       # reuses the bucket
    assert unused.grad is None
    for p in model.parameters():
        p.grad /= 2
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


def _last_json(out):
    import json
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out
    return json.loads(lines[-1])


def test_bench_spawns_its_own_ranks_gloo_dry_run():
    """`python bench.py --gpus 2` starts two ranks itself (torch.distributed.run underneath) and rank 0 prints ONE line with
    n_gpus = 2; the driver's explicit torchrun form runs the same code.  --dry-run --backend gloo: everything but the kernels."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(repo, "bench.py")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["dry_run"] is True and line["scaling"] == "weak"
    assert line["ms_per_step"] >= 1.9          # rank 1 sleeps 2 ms per step: the line carries the MAX over ranks
    assert sum(1 for l in r.stdout.splitlines() if l.startswith("{")) == 1
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), bench, "--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 2
    # a launcher / flag mismatch is refused instead of silently reporting the wrong n_gpus
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), bench, "--gpus", "4", "--dry-run", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0


def test_counting_cost_proxy():
    import numpy as np
    from gsn_amd import dist as gdist, synth
    b = synth.zinc_shape_batch(6, seed=3)
    c = gdist.counting_cost(b.edge_index, b.edge_ptr, 5)
    for g in range(6):
        n, ei = b.graph(g)
        deg = np.bincount(ei[0], minlength=n).astype(np.float64)
        assert abs((deg ** 4).sum() + 1 - c[g]) < 1e-6


def _train_worker(rank, world, port, q):
    """the only parity statement of the data-parallel step is "n_gpu = 1 equals the reference step" -- here the other way
    round: two ranks with half the batch each, averaged gradients, take the same steps as ONE process on the whole batch (a model
    without BatchNorm: its statistics stay per replica by design).  Ten steps; the has-gradient flags ride the collective and are read in
    EVERY call (skipping the exchange let replicas diverge when the set of used parameters grew); one parameter never gets a
    gradient, another one gets its first gradient at step 5 and on rank 1's shard only (a data-dependent branch)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gsn_amd import dist as gdist
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1)).double()
    unused = torch.nn.Parameter(torch.ones(3, dtype=torch.float64))
    late = torch.nn.Parameter(torch.full((1,), 0.5, dtype=torch.float64))
    params = list(model.parameters()) + [unused, late]
    opt = torch.optim.SGD(params, lr=0.05, momentum=0.9, weight_decay=1e-3)
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(64, 6, generator=g, dtype=torch.float64), torch.randn(64, 1, generator=g, dtype=torch.float64)
    lo, hi = gdist.shard_range(64, rank, world)
    reads = []
    for step in range(10):
        opt.zero_grad(set_to_none=True)
        # per-rank MEAN over its shard of equal size: the average over ranks is the whole batch's mean (train_test_funcs.py:88-106)
        out = model(x[lo:hi])
        if step >= 5 and lo >= 32:                       # rows 32 .. 63 only, from the sixth step on
            out = out + late * x[lo:hi, :1]
        torch.nn.functional.mse_loss(out, y[lo:hi]).backward()
        gdist.allreduce_gradients(params, average=True)
        assert (late.grad is not None) == (step >= 5), (rank, step)          # on BOTH ranks, in the very step it appears
        b = list(gdist._BUCKETS.values())[0]
        reads.append(b.stable)
        opt.step()
    assert unused.grad is None and torch.equal(unused.detach(), torch.ones(3, dtype=torch.float64))
    assert reads == [1, 2, 3, 4, 5, 1, 2, 3, 4, 5], reads                  # (read in every call; the set changed once)
    q.put((rank, [p.detach().clone().numpy() for p in list(model.parameters()) + [late]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_take_the_steps_of_one_process_on_the_whole_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1)).double()
    late = torch.nn.Parameter(torch.full((1,), 0.5, dtype=torch.float64))
    opt = torch.optim.SGD(list(model.parameters()) + [late], lr=0.05, momentum=0.9, weight_decay=1e-3)
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(64, 6, generator=g, dtype=torch.float64), torch.randn(64, 1, generator=g, dtype=torch.float64)
    for step in range(10):
        opt.zero_grad(set_to_none=True)
        out = model(x)
        if step >= 5:
            out = out + late * x[:, :1] * (torch.arange(64) >= 32).to(torch.float64).unsqueeze(1)
        torch.nn.functional.mse_loss(out, y).backward()
        opt.step()
    for (r, ps) in res:
        for a, p in zip(ps, list(model.parameters()) + [late]):
            assert abs(a - p.detach().numpy()).max() < 1e-12, r
    for a, b in zip(res[0][1], res[1][1]):
        assert abs(a - b).max() == 0                    # the replicas are identical bit for bit

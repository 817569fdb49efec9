"""The RCCL path inside the GPU suite (a gpurun box has ONE GPU, so world size 1 through the real launcher): process-group
initialisation over RCCL, the flat-bucket gradient all-reduce (forced at world size 1), the same collective as a node of a captured
training step, and bench.py / the training-step script under torch.distributed.run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port(preferred):
    """`preferred` when nothing listens there, else a port the kernel hands out (a rendezvous left in TIME_WAIT by an earlier launcher)."""
    for port in (preferred, 0):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            try:
                s.bind(("127.0.0.1", port))
                return s.getsockname()[1]
            except OSError:
                continue
    return preferred


def _launch(script_args, port, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = _free_port(port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:                   # (kept for the post-mortem: gpurun merges gpurun_out/ back)
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", "launcher_failure_%d.log" % port), "w") as f:
            f.write(" ".join(cmd) + "\n" + r.stdout + "\n" + r.stderr)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    return json.loads(lines[-1])


NO_CACHE = pytest.mark.skipif(os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") == "1",
                              reason="stream capture cannot free memory without the caching allocator (scripts/oob_check.sh)")


@NO_CACHE
def test_allreduce_gradients_through_rccl_and_inside_a_captured_step():
    d = _launch(["scripts/dist_selfcheck.py"], 29541)
    assert d["world"] == 1 and d["rccl_version"]
    assert d["grads_equal_mean_over_ranks"] and d["untouched_parameters_keep_none"]
    assert d["note"] is None, d["note"]
    assert d["replicas_identical_after_replays"] and d["loss_finite"]


def test_training_step_script_under_the_launcher():
    d = _launch(["scripts/train_step_molhiv.py", "--batch", "32", "--steps", "2", "--warmup", "1"], 29542)
    assert d["n_gpus"] == 1 and d["ms_per_step"] > 0 and d["loss"] == d["loss"]


def test_bench_under_the_launcher():
    nog = ["--no-graph"] if os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") == "1" else []      # (scripts/oob_check.sh: no captures without the caching allocator)
    d = _launch(["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--graphs", "4096"] + nog, 29543)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["checked"]["counts_bit_exact"]
    assert d["kernels"]["rccl"]["version"] and d["kernels"]["rccl"]["allreduce_ms"] > 0
    assert len(d["kernels"]["ms_per_step_by_rank"]) == 1 and len(d["kernels"]["graphs_by_rank"]) == 1

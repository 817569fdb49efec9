#!/bin/bash
# new kernels of this session under the out-of-bounds detector (caching allocator off), then the distributed entry points on RCCL with one rank
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3t
timeout 600 python -m pytest tests/test_eval_grad_gpu.py -q -m gpu 2>&1 | tail -2
export PYTORCH_NO_CUDA_MEMORY_CACHING=1
for f in tests/test_eval_grad_gpu.py tests/test_layers_gpu.py tests/test_model_gpu.py tests/test_big_batch_gpu.py; do
  r=$(timeout 900 python -m pytest "$f" -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120); echo "oob $f: $r"
done
r=$(timeout 600 python scripts/train_step_molhiv.py --batch 512 --steps 2 2>&1 | tail -1 | cut -c1-80); echo "oob molhiv step: $r"
unset PYTORCH_NO_CUDA_MEMORY_CACHING
echo "torchrun world 1, nccl:"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>&1 | tail -1 | cut -c1-250
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 scripts/train_step_molhiv.py --batch 4096 --steps 5 2>&1 | tail -1 | cut -c1-250

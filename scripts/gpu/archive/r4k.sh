#!/bin/bash
# transposed-view staging of the 32-row dense kernel: tests, replayed steps
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4k
timeout 1200 python -m pytest tests/test_layers_gpu.py tests/test_linear_small_gpu.py tests/test_graphed_train_gpu.py tests/test_model_gpu.py tests/test_eval_grad_gpu.py tests/test_big_batch_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r4k/test.log
cat gpurun_out/r4k/test.log
for i in 1 2; do
  timeout 300 python scripts/train_step_molhiv.py --batch 32 --steps 100 --warmup 5 --graph 2>&1 | tail -1 | cut -c1-200
  timeout 300 python scripts/train_step_zinc.py --batch 128 --steps 100 --warmup 5 --graph 2>&1 | tail -1 | cut -c1-200
done | tee gpurun_out/r4k/steps.log
timeout 600 python scripts/soak_dense.py 2>&1 | tail -3

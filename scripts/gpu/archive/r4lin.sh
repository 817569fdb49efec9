#!/bin/bash
# 32-row-tile dense kernel: tests, A/B of the small-batch training steps
mkdir -p gpurun_out/r4lin
timeout 900 python -m pytest tests/test_linear_small_gpu.py tests/test_chain_fuzz_gpu.py tests/test_layers_gpu.py tests/test_graphed_train_gpu.py tests/test_fold_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r4lin/test.log
tail -3 gpurun_out/r4lin/test.log
for sm in 0 32 96 192; do
  echo "GSN_LINEAR_SMALL_MAX=$sm"
  GSN_LINEAR_SMALL_MAX=$sm python scripts/train_step_zinc.py --batch 128 --steps 200 --warmup 30 --graph 2>&1 | tail -1 | cut -c150-330
  GSN_LINEAR_SMALL_MAX=$sm python scripts/train_step_molhiv.py --batch 32 --steps 200 --warmup 30 --graph 2>&1 | tail -1 | cut -c150-330
  GSN_LINEAR_SMALL_MAX=$sm python scripts/train_step_zinc.py --batch 4096 --steps 100 --warmup 30 --graph 2>&1 | tail -1 | cut -c150-330
done

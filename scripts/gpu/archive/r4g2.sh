#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4g
timeout 900 python -m pytest tests/test_graphed_train_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r4g/test.log
cat gpurun_out/r4g/test.log | cut -c1-300
for g in "" "--graph"; do
  timeout 300 python scripts/train_step_molhiv.py --batch 32 --steps 50 --warmup 5 $g 2>&1 | tail -1 | cut -c1-330
  timeout 300 python scripts/train_step_zinc.py --batch 128 --steps 50 --warmup 5 $g 2>&1 | tail -1 | cut -c1-330
done | tee gpurun_out/r4g/steps.log
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5

#!/bin/bash
# round 5, item 4: train-mode dense stages on the fp16x3 kernel with the statistics in its epilogue: tests, config-4 step, kernel stats
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5h
(timeout 900 python -m pytest tests/test_chain_fuzz_gpu.py tests/test_big_batch_gpu.py tests/test_model_gpu.py tests/test_layers_gpu.py tests/test_graphed_train_gpu.py tests/test_linear_small_gpu.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r5h/tests.log
cat gpurun_out/r5h/tests.log
(timeout 300 python scripts/gpu/stats_ab.py 2>&1 | tail -10) > gpurun_out/r5h/stats_ab.log
cat gpurun_out/r5h/stats_ab.log
for m in 0 1; do
  echo "GSN_LINEAR_F16X3_STATS=$m"
  GSN_LINEAR_F16X3_STATS=$m timeout 600 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>/dev/null | tail -1 | cut -c1-260
done > gpurun_out/r5h/steps.log
cat gpurun_out/r5h/steps.log
bash scripts/gpu/molhiv_prof.sh > gpurun_out/r5h/molhiv.log 2>&1
head -16 gpurun_out/r5h/molhiv.log | cut -c1-200

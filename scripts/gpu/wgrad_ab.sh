#!/bin/bash
# A/B of the weight-gradient kernels (bf16x6 vs fp32 MFMA) + the tests that cover the native adjoints
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/train
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_layers_gpu.py tests/test_big_batch_gpu.py tests/test_eval_grad_gpu.py -x -q 2>&1 | tail -4 | tee gpurun_out/train/tests.log
echo "bf16x6:"; timeout 600 python scripts/train_step_molhiv.py --batch 4096 --steps 10 2>/dev/null | tail -1 | tee gpurun_out/train/molhiv.json
echo "fp32:"; GSN_WGRAD_FP32=1 timeout 600 python scripts/train_step_molhiv.py --batch 4096 --steps 10 2>/dev/null | tail -1 | tee gpurun_out/train/molhiv_fp32wgrad.json
timeout 600 python scripts/train_step_zinc.py --batch 4096 --steps 10 2>/dev/null | tail -1 | tee gpurun_out/train/zinc.json

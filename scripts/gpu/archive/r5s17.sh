#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s17
(timeout 900 python -m pytest tests/test_fold_self_gpu.py tests/test_fused_encoders_gpu.py tests/test_layers_gpu.py tests/test_eval_grad_gpu.py tests/test_big_batch_gpu.py tests/test_model_gpu.py tests/test_graphed_train_gpu.py -q -x --tb=short 2>&1 | tail -8) > gpurun_out/r5s17/tests.txt
cat gpurun_out/r5s17/tests.txt
{
for f in 0 1 0 1; do
  echo "== FOLD_SELF_ADJOINT=$f"
  GSN_FOLD_SELF_ADJOINT=$f timeout 300 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  GSN_FOLD_SELF_ADJOINT=$f timeout 300 python scripts/train_step_molhiv.py --batch 32 --steps 300 --warmup 3 --graph 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
} > gpurun_out/r5s17/ab.txt 2>&1
cat gpurun_out/r5s17/ab.txt

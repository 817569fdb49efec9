#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s5
{
echo "== d300 b48 train"; timeout 200 python scripts/gpu/diag_fused_enc.py 300 48 train 2>&1 | grep -v amdgpu.ids | head -60
echo "== d300 b48 train, generic kernels"; GSN_PROP_RS=0 GSN_PROP_BWD_PIPE=0 timeout 200 python scripts/gpu/diag_fused_enc.py 300 48 train 2>&1 | grep -v amdgpu.ids | head -60
echo "== d300 b48 eval"; timeout 200 python scripts/gpu/diag_fused_enc.py 300 48 eval 2>&1 | grep -v amdgpu.ids | head -40
} > gpurun_out/r5s5/diag.txt 2>&1
cat gpurun_out/r5s5/diag.txt
(timeout 900 python -m pytest tests/test_fused_encoders_gpu.py tests/test_layers_gpu.py tests/test_eval_grad_gpu.py tests/test_model_gpu.py tests/test_graphed_train_gpu.py tests/test_big_batch_gpu.py -q --tb=line 2>&1 | tail -15) > gpurun_out/r5s5/tests.txt
cat gpurun_out/r5s5/tests.txt

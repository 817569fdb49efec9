#!/bin/bash
# fused edge encoders of GNN_OGB: tests, config-4 step A/B, small-batch replays
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s4
(timeout 900 python -m pytest tests/test_fused_encoders_gpu.py tests/test_layers_gpu.py tests/test_eval_grad_gpu.py tests/test_model_gpu.py tests/test_graphed_train_gpu.py tests/test_big_batch_gpu.py -q -x --tb=short 2>&1 | tail -25) > gpurun_out/r5s4/tests.txt
cat gpurun_out/r5s4/tests.txt
if false; then {
for f in 0 1 0 1; do
  echo "== FUSE_EDGE_ENCODERS=$f"
  GSN_FUSE_EDGE_ENCODERS=$f timeout 300 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>&1 | tail -1 | cut -c150-330
done
for f in 0 1; do
  echo "== graph replay B=32 FUSE=$f"
  GSN_FUSE_EDGE_ENCODERS=$f timeout 300 python scripts/train_step_molhiv.py --batch 32 --steps 200 --warmup 3 --graph 2>&1 | tail -1 | cut -c150-330
done
} > gpurun_out/r5s4/config4b.txt 2>&1; fi


#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/train
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_layers_gpu.py tests/test_big_batch_gpu.py tests/test_range_gpu.py tests/test_encoding_gpu.py -x -q 2>&1 | tail -4 | tee gpurun_out/train/tests.log
timeout 600 python scripts/train_step_molhiv.py --batch 4096 --steps 10 2>/dev/null | tail -1 | tee gpurun_out/train/molhiv.json
timeout 600 python scripts/train_step_zinc.py --batch 4096 --steps 10 2>/dev/null | tail -1 | tee gpurun_out/train/zinc.json

#!/bin/bash
# round 5: GNNSubstructures eval forward hands layer 0 its integer codes (packs, csrc/layer_rp.hip): model tests, full-model step A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5q
(timeout 900 python -m pytest tests/test_model_gpu.py tests/test_big_batch_gpu.py tests/test_end_to_end_gpu.py tests/test_pack16_gpu.py tests/test_codes_gpu.py tests/test_graphed_train_gpu.py -q -m gpu --tb=short 2>&1 | tail -25) | tee gpurun_out/r5q/tests.log | cut -c1-300
for v in 0 1 0 1; do
  echo "GSN_LAYER_PACK16=$v"; GSN_LAYER_PACK16=$v timeout 300 python scripts/profile_full_model.py 2>&1 | tail -1 | cut -c1-300
done | tee gpurun_out/r5q/ab.log

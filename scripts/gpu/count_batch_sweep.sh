#!/bin/bash
# counting kernel: tests, then the pool's pull batch (GSN_PULL_BATCH) against kernel time, then the phase profile
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c
timeout 1500 python -m pytest tests/test_count_gpu.py tests/test_directed_gpu.py tests/test_dataset_gpu.py tests/test_big_batch_gpu.py -x -q 2>&1 | tail -4 | tee gpurun_out/c/ctests.log
for pb in 1 4 8 16 24 1; do
  echo "pull batch $pb: $(GSN_PULL_BATCH=$pb timeout 300 python scripts/gpu/count_ab.py 2>&1 | grep 'int64 rows True  pack True' | tr '\n' ' ')" | tee -a gpurun_out/c/pull_batch.log
done
bash scripts/gpu/cprof.sh 2>&1 | cut -c1-300

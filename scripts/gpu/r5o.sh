#!/bin/bash
# round 5: class indices from the staged 16-bit counts (no byte array: LDS per workgroup 7.5 -> 6.9 KiB): counting tests, the kernel alone A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5o
(timeout 900 python -m pytest tests/test_count_gpu.py tests/test_directed_gpu.py tests/test_dataset_gpu.py tests/test_encoding_gpu.py tests/test_pack16_gpu.py tests/test_codes_gpu.py tests/test_big_batch_gpu.py -q -m gpu --tb=line 2>&1 | tail -4) | tee gpurun_out/r5o/tests.log | cut -c1-300
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export GSN_COUNT_ENC_BYTES=1; else unset GSN_COUNT_ENC_BYTES; fi
  echo "ENC_BYTES=$v"; timeout 300 python scripts/gpu/count_ab.py 2>&1 | grep -E "int64 rows True  pack True|lds" | tail -3
done | tee gpurun_out/r5o/ab.log

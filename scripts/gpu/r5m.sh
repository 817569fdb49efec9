#!/bin/bash
# round 5: (a) counting tests + bench after "workgroups zero their own status words" and "one-hot packs: only the groups a call writes";
# (b) wgrad: workgroups per call (slabs ending in 128 x 128 float atomics) on the config-4 step
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5m
(timeout 900 python -m pytest tests/test_count_gpu.py tests/test_directed_gpu.py tests/test_dataset_gpu.py tests/test_encoding_gpu.py tests/test_pack16_gpu.py tests/test_codes_gpu.py tests/test_packs_gpu.py -q -m gpu --tb=line 2>&1 | tail -5) | tee gpurun_out/r5m/tests.log | cut -c1-300
(timeout 900 python bench.py --no-cpu-baseline 2>gpurun_out/r5m/bench.err | tail -1) > gpurun_out/r5m/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5m/bench.json').read())
k=d['kernels']
print(d['value'], d['ms_per_step'], k['ms_per_step_by_kernel'], 'prepacked', k['step_prepacked']['ms_per_step'], 'no int64', k['step_without_int64_ids']['ms_per_step'], 'zinc12k', k['zinc12k_step']['ms_per_step'], 'small', k['small_batch'])
PY
for w in 512 1024 2048 4096 8192; do
  echo "GSN_WGRAD_WGS=$w"; GSN_WGRAD_WGS=$w timeout 600 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>/dev/null | tail -1 | cut -c120-200
done | tee gpurun_out/r5m/wgrad.log

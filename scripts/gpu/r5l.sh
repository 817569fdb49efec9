#!/bin/bash
# round 5: counting kernel with 16-bit staged counts: counting tests (bit-exact), then the bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5l
(timeout 1200 python -m pytest tests/test_count_gpu.py tests/test_directed_gpu.py tests/test_dataset_gpu.py tests/test_big_batch_gpu.py tests/test_encoding_gpu.py tests/test_pack16_gpu.py -q -m gpu --tb=line 2>&1 | tail -12) > gpurun_out/r5l/tests.log
cat gpurun_out/r5l/tests.log | cut -c1-600
(timeout 900 python bench.py 2>gpurun_out/r5l/bench.err | tail -1) > gpurun_out/r5l/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5l/bench.json').read())
k=d['kernels']
print(d['value'], d['ms_per_step'], k['ms_per_step_by_kernel'], 'prepacked', k['step_prepacked']['ms_per_step'], 'no int64', k['step_without_int64_ids']['ms_per_step'], 'zinc12k', k['zinc12k_step']['ms_per_step'], 'er128', k['count_er128_config5']['graphs_per_s'], 'full', k['full_model_step']['ms_per_step'], 'cfg4', k['train_step_config4']['ms_per_step'])
print(d['checked'])
PY
tail -3 gpurun_out/r5l/bench.err

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s14
(timeout 900 python -m pytest tests/test_count_pack_only_gpu.py tests/test_count_gpu.py tests/test_pack16_gpu.py tests/test_codes_gpu.py tests/test_dataset_gpu.py -q -x --tb=short 2>&1 | tail -12) > gpurun_out/r5s14/tests.txt
cat gpurun_out/r5s14/tests.txt
{
for f in 0 1 0 1; do echo "== PACK_ONLY_IDS=$f"; GSN_BENCH_PACK_ONLY_IDS=$f timeout 300 python bench.py --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']
print(d['value'], d['ms_per_step'], k.get('ms_per_step_by_kernel'), d['checked'].get('encoded_rows_equal_one_hot_of_counts'), d['checked'].get('int64_identifiers_of_the_timed_step_equal_a_plain_counting_launch'), d['checked']['layer_elementwise_1e-5_rel_plus_1e-5_rowmax'])"; done
} > gpurun_out/r5s14/bench_ab.txt 2>&1
cat gpurun_out/r5s14/bench_ab.txt

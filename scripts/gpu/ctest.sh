#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c
timeout 1500 python -m pytest tests/test_count_gpu.py tests/test_directed_gpu.py tests/test_dataset_gpu.py -x -q 2>&1 | tail -8 | tee gpurun_out/c/ctests.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels']['ms_per_step_by_kernel'], d['counts_checked'], d['checked'])" | tee gpurun_out/c/bench_pair.log
GSN_COUNT_PAIR=0 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels']['ms_per_step_by_kernel'])" | tee gpurun_out/c/bench_nopair.log

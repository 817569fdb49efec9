#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s13
(timeout 900 python -m pytest tests/test_chain_fuzz_gpu.py tests/test_range_gpu.py tests/test_big_batch_gpu.py -q -x --tb=short 2>&1 | tail -4) > gpurun_out/r5s13/tests.txt
cat gpurun_out/r5s13/tests.txt
{
for w in 0 512 1024 2048 0 1024; do
  echo "== L16_SPLIT_WGS=$w"
  GSN_L16_SPLIT_WGS=$w timeout 300 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
} > gpurun_out/r5s13/ab.txt 2>&1
cat gpurun_out/r5s13/ab.txt
cd /tmp && export TMPDIR=/tmp
for w in 0 1024; do
GSN_L16_SPLIT_WGS=$w timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5s13 -o s$w -- python $GRAFT_REPO_ROOT/scripts/train_step_molhiv.py --batch 4096 --steps 5 --warmup 2 > /dev/null 2>&1
grep -h "lin16_split" $(find $GRAFT_REPO_ROOT/gpurun_out/r5s13 -name "s${w}_kernel_stats.csv") | cut -c1-200
done

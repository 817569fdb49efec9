#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s15
(timeout 900 python -m pytest tests/test_encoding_gpu.py tests/test_fused_encoders_gpu.py tests/test_model_gpu.py tests/test_graphed_train_gpu.py -q -x --tb=short 2>&1 | tail -5) > gpurun_out/r5s15/tests.txt
cat gpurun_out/r5s15/tests.txt
{
for f in 0 1 0 1; do
  echo "== EMBED_PIPE=$f"
  GSN_EMBED_PIPE=$f timeout 300 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
} > gpurun_out/r5s15/ab.txt 2>&1
cat gpurun_out/r5s15/ab.txt
cd /tmp && export TMPDIR=/tmp
for w in 0 1; do
GSN_EMBED_PIPE=$w timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5s15 -o s$w -- python $GRAFT_REPO_ROOT/scripts/train_step_molhiv.py --batch 4096 --steps 5 --warmup 2 > /dev/null 2>&1
grep -h "embed_lds" $(find $GRAFT_REPO_ROOT/gpurun_out/r5s15 -name "s${w}_kernel_stats.csv") | cut -c1-160
done

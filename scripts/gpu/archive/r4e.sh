#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for v in 8192 1000000000 8192 1000000000; do
  echo "GSN_EMBED_LDS_MIN_ROWS=$v"; GSN_EMBED_LDS_MIN_ROWS=$v timeout 600 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>&1 | tail -1 | cut -c150-215
done
cd /tmp && export TMPDIR=/tmp
GSN_EMBED_LDS_MIN_ROWS=1000000000 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/molhiv2 -o m -- python $ROOT/scripts/train_step_molhiv.py --batch 4096 --steps 10 > $ROOT/gpurun_out/molhiv2/m.log 2>&1 </dev/null
grep -h "embed" $ROOT/gpurun_out/molhiv2/*/m_kernel_stats.csv $ROOT/gpurun_out/molhiv2/m_kernel_stats.csv 2>/dev/null | cut -c1-160

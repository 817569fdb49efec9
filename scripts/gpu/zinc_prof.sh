#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/zincprof
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/zincprof -o z -- python $ROOT/scripts/train_step_zinc.py --batch 4096 --steps 10 > $ROOT/gpurun_out/zincprof/z.log 2>&1 </dev/null
tail -1 $ROOT/gpurun_out/zincprof/z.log | cut -c1-300

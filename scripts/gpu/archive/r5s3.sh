#!/bin/bash
# pipelined propagate kernels: config-4 step A/B (all off / forward only / all on), stand-alone fractions, tests
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s3
{
for cfg in "0 0 0" "1 1 0" "1 1 1" "0 0 0" "1 1 1"; do
  set -- $cfg
  rs=""; cp=""; [ $1 = 0 ] && rs="GSN_PROP_RS=0"; [ $2 = 0 ] && cp="GSN_PROP_CP=0"
  echo "== RS=$1 CP=$2 BWD=$3"
  env $rs $cp GSN_PROP_BWD_PIPE=$3 timeout 300 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>&1 | tail -1 | cut -c1-330
done
} > gpurun_out/r5s3/config4.txt 2>&1
cat gpurun_out/r5s3/config4.txt
(timeout 300 python scripts/bench_propagate.py 2>&1 | tail -1) > gpurun_out/r5s3/bench_propagate.txt
cat gpurun_out/r5s3/bench_propagate.txt
(timeout 1500 python -m pytest tests -q -m gpu -x --tb=short -k "propag or csr or pool or readout or gin or ogb or big_batch or layers_gpu or model or eval_grad or graphed" 2>&1 | tail -8) > gpurun_out/r5s3/tests.txt
cat gpurun_out/r5s3/tests.txt

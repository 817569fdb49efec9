#!/bin/bash
# round 5, item 4: the 5 x 300 golden step under the dense-stage modes (digest deviation), the config-4 step time per mode, its kernel stats
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5e
(timeout 600 python scripts/gpu/ogb300_modes.py 2>&1 | grep -E "^mode|Error|error" ) > gpurun_out/r5e/modes.log
cat gpurun_out/r5e/modes.log
for m in 0 1; do
  echo "GSN_LINEAR_F16X3_STATS=$m"
  GSN_LINEAR_F16X3_STATS=$m timeout 600 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>/dev/null | tail -1 | cut -c1-260
done > gpurun_out/r5e/steps.log
cat gpurun_out/r5e/steps.log
bash scripts/gpu/molhiv_prof.sh > gpurun_out/r5e/molhiv.log 2>&1
head -24 gpurun_out/r5e/molhiv.log | cut -c1-200

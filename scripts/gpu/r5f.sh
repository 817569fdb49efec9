#!/bin/bash
# round 5: (a) digest deviation of the 5 x 300 golden step with every dense stage forced onto the fp16x3 kernel, (b) kernel stats of the full-model step
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
mkdir -p gpurun_out/r5f
(timeout 600 python scripts/gpu/ogb300_modes.py "f16x3_all:GSN_LINEAR_F16X3_STATS=1,GSN_LINEAR_F16X3_MIN_TILES=0" "f16x3_fwd_only:GSN_LINEAR_F16X3_MIN_TILES=0" 2>&1 | grep -E "^mode|Error|error" ) > gpurun_out/r5f/modes.log
cat gpurun_out/r5f/modes.log
(timeout 300 python scripts/profile_full_model.py 2>&1 | tail -1) > gpurun_out/r5f/full_model.json
cat gpurun_out/r5f/full_model.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r5f -o fm -- python $ROOT/scripts/profile_full_model.py > $ROOT/gpurun_out/r5f/fm.log 2>&1 </dev/null
python - <<PY
import csv,glob
f=glob.glob("$ROOT/gpurun_out/r5f/**/fm_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:24]: print("%-100s calls %5s avg %9.1f us  %5s%%" % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage'][:5]))
PY

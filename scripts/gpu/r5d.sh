#!/bin/bash
# full-model step: family timers + rocprof kernel stats (16 384 graphs)
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5d; mkdir -p $OUT
timeout 300 python scripts/profile_full_model.py 2>&1 | tail -1 | tee $OUT/fm.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fm -- python $ROOT/scripts/profile_full_model.py > $OUT/fm.log 2>&1 </dev/null
cd $ROOT
f=$(find $OUT -name "fm_kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/fm_kernels.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print("%-90s calls %5s avg %9.1f us total %6.2f%%"%(r["Name"][:90],r["Calls"],float(r["AverageNs"])/1e3,float(r["Percentage"])))
PY

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/zinc4096
cd /tmp && export TMPDIR=/tmp
timeout 600 python $ROOT/scripts/train_step_zinc.py --batch 4096 --steps 20 --warmup 10 2>/dev/null | tail -1 | cut -c1-260
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/zinc4096 -o z -- python $ROOT/scripts/train_step_zinc.py --batch 4096 --steps 10 > $ROOT/gpurun_out/zinc4096/z.log 2>&1 </dev/null
python - <<PY
import csv,glob
f=glob.glob("$ROOT/gpurun_out/zinc4096/**/z_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("kernel ms per step", tot/13/1e6)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:24]: print("%-95s calls %5s avg %9.1f us  %5.2f%%" % (r['Name'][:95], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/molhiv
cd /tmp && export TMPDIR=/tmp
timeout 600 python $ROOT/scripts/train_step_molhiv.py --batch 4096 --steps 10 2>/dev/null | tail -1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/molhiv -o m -- python $ROOT/scripts/train_step_molhiv.py --batch 4096 --steps 10 > $ROOT/gpurun_out/molhiv/m.log 2>&1 </dev/null
python - <<PY
import csv,glob
f=glob.glob("$ROOT/gpurun_out/molhiv/**/m_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]: print("%-95s calls %5s avg %9.1f us  %5s%%" % (r['Name'][:95], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage'][:5]))
PY

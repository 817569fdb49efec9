#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r4fm
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r4fm -o fm -- python $ROOT/scripts/profile_full_model.py > $ROOT/gpurun_out/r4fm/fm.log 2>&1 </dev/null
cd $ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4fm/**/fm_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%-95s %6s calls %8.1f us avg %6.2f%%" % (r["Name"][:95], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -2 gpurun_out/r4fm/fm.log | cut -c1-300

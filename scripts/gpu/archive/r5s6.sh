#!/bin/bash
# kernel stats of the replayed small-batch steps at HEAD
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r5s6
cd /tmp && export TMPDIR=/tmp
for w in zinc molhiv; do
  b=128; [ $w = molhiv ] && b=32
  timeout 300 python $ROOT/scripts/train_step_$w.py --batch $b --steps 300 --warmup 3 --graph 2>/dev/null | tail -1 | cut -c1-60,140-260
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r5s6 -o $w -- python $ROOT/scripts/train_step_$w.py --batch $b --steps 100 --warmup 3 --graph > $ROOT/gpurun_out/r5s6/prof_$w.log 2>&1 </dev/null
done
cd $ROOT
python - <<'PY'
import csv, glob
for w in ("zinc", "molhiv"):
    f = glob.glob("gpurun_out/r5s6/**/%s_kernel_stats.csv" % w, recursive=True)
    if not f: print(w, "no stats"); continue
    rows = list(csv.DictReader(open(f[0])))
    calls = sum(int(r["Calls"]) for r in rows); tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(w, "kernels", len(rows), "calls", calls, "total ms", tot / 1e6, "(per step %.1f calls, %.3f ms)" % (calls / 104, tot / 1e6 / 104))
    for r in rows[:40]:
        print("  %-100s %6.1f/step %7.2f us %5.1f%%" % (r["Name"][:100], int(r["Calls"]) / 104, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY

#!/bin/bash
# round 5: the whole GPU suite + smoke at the train-mode fp16x3 change, then kernel stats of the replayed small-batch steps
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
mkdir -p gpurun_out/r5j
(timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6) > gpurun_out/r5j/all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r5j/smoke.log
cat gpurun_out/r5j/all.log gpurun_out/r5j/smoke.log
cd /tmp && export TMPDIR=/tmp
for w in zinc molhiv; do
  b=128; [ $w = molhiv ] && b=32
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r5j -o $w -- python $ROOT/scripts/train_step_$w.py --batch $b --steps 100 --warmup 3 --graph > $ROOT/gpurun_out/r5j/prof_$w.log 2>&1 </dev/null
done
cd $ROOT
python - <<'PY'
import csv, glob
for w in ("zinc", "molhiv"):
    f = glob.glob("gpurun_out/r5j/**/%s_kernel_stats.csv" % w, recursive=True)
    if not f: print(w, "no stats"); continue
    rows = list(csv.DictReader(open(f[0])))
    calls = sum(int(r["Calls"]) for r in rows); tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(w, "kernels", len(rows), "calls", calls, "total ms", tot / 1e6, "(104 steps incl. warm-up: per step %.1f calls, %.3f ms)" % (calls / 104, tot / 1e6 / 104))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
        print("   %-90s %6s calls %8.1f us avg %6.2f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY

#!/bin/bash
# whole-step HIP graph: tests, eager vs replay at the reference's batch sizes, kernel stats of the replayed steps
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
mkdir -p gpurun_out/r4g
timeout 900 python -m pytest tests/test_graphed_train_gpu.py -x -q 2>&1 | tail -40 > gpurun_out/r4g/test.log
cat gpurun_out/r4g/test.log
for g in "" "--graph"; do
  timeout 300 python scripts/train_step_molhiv.py --batch 32 --steps 50 --warmup 5 $g 2>&1 | tail -1 | cut -c1-330
  timeout 300 python scripts/train_step_zinc.py --batch 128 --steps 50 --warmup 5 $g 2>&1 | tail -1 | cut -c1-330
done | tee gpurun_out/r4g/steps.log
cd /tmp && export TMPDIR=/tmp
for w in zinc molhiv; do
  b=128; [ $w = molhiv ] && b=32
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r4g -o $w -- python $ROOT/scripts/train_step_$w.py --batch $b --steps 100 --warmup 3 --graph > $ROOT/gpurun_out/r4g/prof_$w.log 2>&1 </dev/null
done
cd $ROOT
python - <<'PY'
import csv, glob
for w in ("zinc", "molhiv"):
    f = glob.glob("gpurun_out/r4g/**/%s_kernel_stats.csv" % w, recursive=True)
    if not f: print(w, "no stats"); continue
    rows = list(csv.DictReader(open(f[0])))
    calls = sum(int(r["Calls"]) for r in rows); tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(w, "kernels", len(rows), "calls", calls, "total ms", tot / 1e6, "(104 steps incl. warm-up: per step %.1f calls, %.3f ms)" % (calls / 104, tot / 1e6 / 104))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
        print("   %-90s %6s calls %8.1f us avg %6.2f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY

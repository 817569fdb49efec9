#!/bin/bash
# round-4 last session: kernel stats of the replayed small-batch training steps at HEAD, then the bench line
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
mkdir -p gpurun_out/r4h
for g in "--graph"; do
  timeout 300 python scripts/train_step_molhiv.py --batch 32 --steps 50 --warmup 5 $g 2>&1 | tail -1 | cut -c1-330
  timeout 300 python scripts/train_step_zinc.py --batch 128 --steps 50 --warmup 5 $g 2>&1 | tail -1 | cut -c1-330
done | tee gpurun_out/r4h/steps.log
cd /tmp && export TMPDIR=/tmp
for w in zinc molhiv; do
  b=128; [ $w = molhiv ] && b=32
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r4h -o $w -- python $ROOT/scripts/train_step_$w.py --batch $b --steps 100 --warmup 3 --graph > $ROOT/gpurun_out/r4h/prof_$w.log 2>&1 </dev/null
done
cd $ROOT
python - <<'PY'
import csv, glob
for w in ("zinc", "molhiv"):
    f = glob.glob("gpurun_out/r4h/**/%s_kernel_stats.csv" % w, recursive=True)
    if not f: print(w, "no stats"); continue
    rows = list(csv.DictReader(open(f[0])))
    calls = sum(int(r["Calls"]) for r in rows); tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(w, "kernels", len(rows), "calls", calls, "total ms", tot / 1e6, "(per step %.1f calls, %.3f ms)" % (calls / 104, tot / 1e6 / 104))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:60]:
        print("   %-90s %6s calls %8.1f us avg %6.2f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
(timeout 900 python bench.py 2> gpurun_out/r4h/bench.err | tail -1) > gpurun_out/r4h/bench.json
cut -c1-600 gpurun_out/r4h/bench.json

#!/bin/bash
# round 5 validation: whole GPU suite + smoke, rocprofv3 passes over bench.py (kernel stats, SQ counters, FETCH / WRITE) + the bench line,
# kernel stats of the config-4 step and of the replayed small-batch steps, layer_g check
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
mkdir -p gpurun_out/r5final
(timeout 1800 python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -15) > gpurun_out/r5final/all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r5final/smoke.log
cat gpurun_out/r5final/all.log gpurun_out/r5final/smoke.log | cut -c1-500
bash scripts/profile_bench.sh r5prof > gpurun_out/r5final/prof.log 2>&1
tail -1 gpurun_out/r5prof/bench.json | cut -c1-400
bash scripts/gpu/molhiv_prof.sh > gpurun_out/r5final/molhiv.log 2>&1
head -3 gpurun_out/r5final/molhiv.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
for w in zinc molhiv; do
  b=128; [ $w = molhiv ] && b=32
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r5final -o $w -- python $ROOT/scripts/train_step_$w.py --batch $b --steps 100 --warmup 3 --graph > $ROOT/gpurun_out/r5final/prof_$w.log 2>&1 </dev/null
done
cd $ROOT
python - <<'PY'
import csv, glob
for w in ("zinc", "molhiv"):
    f = glob.glob("gpurun_out/r5final/**/%s_kernel_stats.csv" % w, recursive=True)
    if not f: print(w, "no stats"); continue
    rows = list(csv.DictReader(open(f[0])))
    calls = sum(int(r["Calls"]) for r in rows); tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(w, "kernels", len(rows), "calls", calls, "total ms", tot / 1e6, "(104 steps incl. warm-up: per step %.1f calls, %.3f ms)" % (calls / 104, tot / 1e6 / 104))
PY
(timeout 300 python scripts/gpu/g_check.py --time 2>&1 | tail -4) > gpurun_out/r5final/gcheck.log
cat gpurun_out/r5final/gcheck.log

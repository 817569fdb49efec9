#!/bin/bash
# round 5 validation: whole GPU suite + smoke, rocprofv3 passes over bench.py (kernel stats, SQ counters, FETCH / WRITE) + the bench line,
# kernel stats of the config-4 step and of the replayed small-batch steps, layer_g check
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
mkdir -p gpurun_out/r5fin2
(timeout 1800 python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -15) > gpurun_out/r5fin2/all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r5fin2/smoke.log
cat gpurun_out/r5fin2/all.log gpurun_out/r5fin2/smoke.log | cut -c1-500
bash scripts/profile_bench.sh r5prof2 > gpurun_out/r5fin2/prof.log 2>&1
tail -1 gpurun_out/r5prof2/bench.json | cut -c1-400
bash scripts/gpu/molhiv_prof.sh > gpurun_out/r5fin2/molhiv.log 2>&1
head -3 gpurun_out/r5fin2/molhiv.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
for w in zinc molhiv; do
  b=128; [ $w = molhiv ] && b=32
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r5fin2 -o $w -- python $ROOT/scripts/train_step_$w.py --batch $b --steps 100 --warmup 3 --graph > $ROOT/gpurun_out/r5fin2/prof_$w.log 2>&1 </dev/null
done
cd $ROOT
python - <<'PY'
import csv, glob
for w in ("zinc", "molhiv"):
    f = glob.glob("gpurun_out/r5fin2/**/%s_kernel_stats.csv" % w, recursive=True)
    if not f: print(w, "no stats"); continue
    rows = list(csv.DictReader(open(f[0])))
    calls = sum(int(r["Calls"]) for r in rows); tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(w, "kernels", len(rows), "calls", calls, "total ms", tot / 1e6, "(104 steps incl. warm-up: per step %.1f calls, %.3f ms)" % (calls / 104, tot / 1e6 / 104))
PY
(timeout 300 python scripts/gpu/g_check.py --time 2>&1 | tail -4) > gpurun_out/r5fin2/gcheck.log
cat gpurun_out/r5fin2/gcheck.log
# FETCH_SIZE / WRITE_SIZE of the stand-alone propagate launches (profiles/r05_propagate_pmc.csv), own passes
cd /tmp && export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out/r5prop
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/r5prop -o f -- python $ROOT/scripts/bench_propagate.py > $ROOT/gpurun_out/r5prop/f.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/r5prop -o w -- python $ROOT/scripts/bench_propagate.py > $ROOT/gpurun_out/r5prop/w.log 2>&1 </dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r5prop -o r -- python $ROOT/scripts/bench_propagate.py > $ROOT/gpurun_out/r5prop/r.log 2>&1 </dev/null
cd $ROOT
tail -1 gpurun_out/r5prop/r.log | cut -c1-900

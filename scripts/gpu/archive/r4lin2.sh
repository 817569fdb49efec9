#!/bin/bash
# molhiv B=32 replay: f16x3 trio vs bf16x6 32-row tiles for the small products (kernel stats of both)
ROOT=$(pwd)
mkdir -p gpurun_out/r4lin
timeout 900 python -m pytest tests/test_linear_small_gpu.py tests/test_chain_fuzz_gpu.py tests/test_layers_gpu.py tests/test_graphed_train_gpu.py tests/test_fold_gpu.py tests/test_eval_grad_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r4lin/test2.log
tail -3 gpurun_out/r4lin/test2.log
cd /tmp && export TMPDIR=/tmp
for t in 0 96; do
  GSN_LINEAR_F16X3_MIN_TILES=$t python $ROOT/scripts/train_step_molhiv.py --batch 32 --steps 200 --warmup 30 --graph 2>&1 | tail -1 | cut -c150-260
  GSN_LINEAR_F16X3_MIN_TILES=$t timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r4lin -o molhiv_t$t -- python $ROOT/scripts/train_step_molhiv.py --batch 32 --steps 100 --warmup 3 --graph > $ROOT/gpurun_out/r4lin/prof_t$t.log 2>&1 </dev/null
  tail -1 $ROOT/gpurun_out/r4lin/prof_t$t.log | cut -c150-260
done
cd $ROOT
python - <<'PY'
import csv, glob
for t in (0, 96):
    f = glob.glob("gpurun_out/r4lin/**/molhiv_t%d_kernel_stats.csv" % t, recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    calls = sum(int(r["Calls"]) for r in rows); tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("MIN_TILES", t, "calls/step %.1f  kernel ms/step %.3f" % (calls / 104, tot / 1e6 / 104))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
        print("   %-100s %6s calls %8.1f us avg %6.2f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY

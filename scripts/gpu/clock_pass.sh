#!/bin/bash
# effective shader clock per kernel of the bench step: GRBM_GUI_ACTIVE / kernel wall time (MI355X_MICROARCH.md, DVFS give-back)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4clk
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT" -o g -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-graph > "$OUT/g.log" 2>&1 </dev/null
cd $ROOT
python - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/r4clk/**/g_counter_collection.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/r4clk/**/g_kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
        continue
    ns, name = dur[r["Dispatch_Id"]]
    acc[name.split("(")[0][-48:]].append((float(r["Counter_Value"]), ns))
print("kernel, dispatches, GRBM_GUI_ACTIVE per dispatch, wall ns per dispatch, GHz (if the counter is one value per dispatch), GHz / 8 (if summed over the 8 XCDs)")
for k, v in sorted(acc.items(), key=lambda kv: -sum(x[1] for x in kv[1]))[:6]:
    c = sum(x[0] for x in v) / len(v); n = sum(x[1] for x in v) / len(v)
    print("%s, %d, %.0f, %.0f, %.3f, %.3f" % (k, len(v), c, n, c / n, c / n / 8))
PY

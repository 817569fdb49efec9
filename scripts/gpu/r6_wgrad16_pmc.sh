#!/bin/bash
# r06: counters of wgrad_f16x3_kernel (and the bf16x6 kernel beside it) at 105 083 x 300 x 600
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6wg
mkdir -p "$OUT"
CMD="python $ROOT/scripts/gpu/r6_wgrad16.py"
cd /tmp && export TMPDIR=/tmp
export R6_WGRAD16_ONE=1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS \
    --kernel-trace --output-format csv -d "$OUT" -o p -- $CMD > "$OUT/p.log" 2>&1 </dev/null
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS \
    --kernel-trace --output-format csv -d "$OUT" -o q -- $CMD > "$OUT/q.log" 2>&1 </dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o f -- $CMD > "$OUT/f.log" 2>&1 </dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o w -- $CMD > "$OUT/w.log" 2>&1 </dev/null
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$OUT" -o t -- $CMD > "$OUT/t.log" 2>&1 </dev/null
cd "$ROOT"
python - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r6wg")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "wgrad" not in k: continue
        acc[k.split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY

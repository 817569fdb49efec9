#!/bin/bash
# layer_g: phase profile (GSN_FUSED_PROF), SQ counters, kernel stats; refill test many times
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5b
mkdir -p $OUT
bash scripts/gpu/flake_refill.sh 12 > $OUT/flake.log 2>&1
sort $OUT/flake.log | uniq -c | cut -c1-300
(GSN_FUSED_PROF=1 timeout 200 python scripts/gpu/g_check.py --time --skip-checks 2>&1 | grep -E "gprof|layer_g") > $OUT/gprof.log
cat $OUT/gprof.log
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/scripts/gpu/g_check.py --time --skip-checks"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT -o p -- $CMD > $OUT/p.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT -o q -- $CMD > $OUT/q.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o c -- $CMD > $OUT/c.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o f -- $CMD > $OUT/f.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o w -- $CMD > $OUT/w.log 2>&1 </dev/null
cd $ROOT
python - $OUT <<'PY' | tee $OUT/pmc_summary.txt
import csv,glob,sys,os
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for p in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    per=defaultdict(float)
    for r in csv.DictReader(open(p)):
        if "layer_fused" not in r["Kernel_Name"] or "prepare" in r["Kernel_Name"]: continue
        per[(r["Kernel_Name"][:40],r["Dispatch_Id"],r["Counter_Name"])]+=float(r["Counter_Value"])
    for (k,d,c),v in per.items(): acc[k][c].append(v)
for k,cs in acc.items():
    print(k)
    for c,v in sorted(cs.items()): print("   %-28s %.4g (n=%d)"%(c,sum(v)/len(v),len(v)))
PY

#!/bin/bash
# counting kernel: issue mix / waits / instruction-cache counters (SQ, SQC), rocprofv3 --pmc passes over scripts/gpu/count_ab.py
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-cissue}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQC?_[A-Z0-9_]+|GRBM_[A-Z0-9_]+" | sort -u > $OUT/counters_available.txt
have() { for c in "$@"; do grep -qx "$c" $OUT/counters_available.txt && echo -n "$c "; done; }
P1=$(have SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS)
P2=$(have SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU)
P3=$(have SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_WAIT_IFETCH)
P4=$(have SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_WAVES GRBM_GUI_ACTIVE)
P5=$(have SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_INPUT_VALID_READYB)
P6=$(have SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_EXP_GDS SQ_INSTS_SENDMSG SQ_INSTS_WAVE32 SQ_ACTIVE_INST_EXP_GDS SQ_INSTS_VSKIPPED)
CMD="python $ROOT/scripts/gpu/count_ab.py"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1)); echo "P$i $P"
  [ -z "$P" ] && continue
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT -o c_p$i -- $CMD > $OUT/c_p$i.log 2>&1 </dev/null
done
cd $ROOT
python - $OUT <<'PY' | tee $OUT/issue.txt
import csv,glob,sys,os
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for p in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    per=defaultdict(float)
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"]
        if "count_kernel" not in k: continue
        per[(k.split("(")[0][:60],r["Dispatch_Id"],r["Counter_Name"])]+=float(r["Counter_Value"])
    for (k,d,c),v in per.items(): acc[k][c].append(v)
for k,cs in sorted(acc.items()):
    print(k)
    for c,v in sorted(cs.items()): print("   %-32s %.5g (n=%d)"%(c,sum(v)/len(v),len(v)))
PY

#!/bin/bash
# which instruction classes occupy the issue slots of the one-launch layer kernels, and what the waves wait for.
# SQ counters in passes of <= 8 (PMC never combined with sys / hip traces), layer_rp through scripts/bench_layer.py (fused_pack16),
# layer_g through scripts/gpu/g_check.py; the table goes to gpurun_out/<tag>/issue.txt (-> profiles/r05_layer_rp_issue.txt)
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-issue}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+|GRBM_[A-Z0-9_]+" | sort -u > $OUT/counters_available.txt
have() { for c in "$@"; do grep -qx "$c" $OUT/counters_available.txt && echo -n "$c "; done; }
P1=$(have SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA)
P2=$(have SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INSTS_LDS)
P3=$(have SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR)
P4=$(have SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM)
P5=$(have GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VSKIPPED)
echo "P1 $P1"; echo "P2 $P2"; echo "P3 $P3"; echo "P4 $P4"; echo "P5 $P5"
for tag in rp g; do
  if [ $tag = rp ]; then CMD="python $ROOT/scripts/bench_layer.py --graphs 65536 --steps 3"; else CMD="python $ROOT/scripts/gpu/g_check.py --time --skip-checks"; fi
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i+1))
    [ -z "$P" ] && continue
    timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT -o ${tag}_p$i -- $CMD > $OUT/${tag}_p$i.log 2>&1 </dev/null
  done
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
        if "layer_fused_kernel" not in k or "prepare" in k: continue
        per[(k.split("(")[0][:60],r["Dispatch_Id"],r["Counter_Name"])]+=float(r["Counter_Value"])
    for (k,d,c),v in per.items(): acc[k][c].append(v)
for k,cs in sorted(acc.items()):
    print(k)
    for c,v in sorted(cs.items()): print("   %-32s %.5g (n=%d)"%(c,sum(v)/len(v),len(v)))
PY

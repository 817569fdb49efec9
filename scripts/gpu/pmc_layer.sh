# SQ counters of the layer kernel over scripts/bench_layer.py (own passes; PMC never combined with sys/hip traces)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/scripts/bench_layer.py --graphs 65536 --steps 3"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT -o p -- $CMD > $OUT/p.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT -o q -- $CMD > $OUT/q.log 2>&1 </dev/null
cd $ROOT
python - $OUT <<'PY'
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

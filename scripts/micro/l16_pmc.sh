#!/bin/bash
# L2 counters of the fp16x3 linear path at one shape (own PMC passes, kernel trace only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/l16c
  GSN_L16_SHAPES=${SHAPES:-0} rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/l16c -o c -- python $ROOT/scripts/bench_linear.py > /tmp/l16c.log 2>&1
  f=$(find /tmp/l16c -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "planes_kernel" in k or "dma_kernel" in k or "split_rows" in k:
        name = "split" if "split_rows" in k else "matrix"
        a = acc[(name, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for (n, c), (v, k) in sorted(acc.items()):
    print("%-8s %-28s %.4g per launch" % (n, c, v / k))
PY
done

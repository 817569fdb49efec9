#!/bin/bash
# per-kernel time of the fp16x3 linear path at one shape under the planes kernel's diagnostic switches (GSN_L16_DBG bits:
# 1 no stores, 2 no products, 4 no loads, 8 no LDS writes; results are garbage with any bit set).  SHAPES: indices of
# bench_linear.py's list or M:K:N.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for d in ${DBGS:-0 1 2 4 8}; do
  rm -rf /tmp/l16p
  GSN_L16_DBG=$d GSN_L16_SHAPES=${SHAPES:-0} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/l16p -o r -- python $ROOT/scripts/bench_linear.py > /tmp/l16.log 2>&1
  f=$(find /tmp/l16p -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$d" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    for key in ("dma_kernel", "planes_kernel", "split_rows", "prepare_kernel", "linear_fwd_bf16"):
        if key in n:
            out.append("%s avg %.1f min %.1f max %.1f us x%s" % (key, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Calls"]))
print("dbg %s: %s" % (sys.argv[2], "  ".join(out)))
PY
done

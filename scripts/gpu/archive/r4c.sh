#!/bin/bash
# layer_rp ablations: what the tile's cycles are sensitive to (results of these builds are garbage)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4e
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_pack16_gpu.py -x -q 2>&1 | tail -3
for v in "" pd8 nt drain pd8nt g0 nostore pd8nostore; do
  lib=$ROOT/gsn_amd/lib/libgsn_hip.so; [ -n "$v" ] && lib=$ROOT/gsn_amd/lib/variants/libgsn_hip_$v.so
  echo "== variant ${v:-product}"
  GSN_LIB_PATH=$lib timeout 300 python scripts/bench_layer.py > "$OUT/layer_${v:-product}.json" 2>"$OUT/layer_${v:-product}.err" </dev/null
  python - "$OUT/layer_${v:-product}.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k]["ms_per_layer"] for k in ("fused_pack16", "fused", "multi_launch") if k in d}, d.get("pack16_max_diff_over_max"))
PY
  GSN_LIB_PATH=$lib GSN_FUSED_PROF=1 timeout 300 python scripts/bench_layer.py --steps 16 > /dev/null 2>"$OUT/prof_${v:-product}.err" </dev/null
  grep -h "rpprof range" "$OUT/prof_${v:-product}.err" | tail -2
done

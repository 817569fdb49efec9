#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for v in "" store4 nostore "" store4 nostore; do
  lib=gsn_amd/lib/libgsn_hip.so; [ -n "$v" ] && lib=gsn_amd/lib/variants/libgsn_hip_$v.so
  GSN_LIB_PATH=$(pwd)/$lib timeout 300 python scripts/bench_layer.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('${v:-product}', {k: d[k]['ms_per_layer'] for k in ('fused_pack16','fused') if k in d})"
done

#!/bin/bash
# layer_rp: share of the nodes given to the older wave of every SIMD (GSN_RP_OLD_SHARE) -> layer time, wave-life spread
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/rpshare
for s in 0.5 0.54 0.58 0.62 0.66 0.5; do
  t=$(GSN_RP_OLD_SHARE=$s timeout 200 python scripts/bench_layer.py --graphs 65536 --steps 40 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['fused_pack16']['kernels_ms'], d['pack16_max_diff_over_max'])")
  p=$(GSN_RP_OLD_SHARE=$s GSN_FUSED_PROF=1 timeout 200 python scripts/bench_layer.py --graphs 65536 --steps 16 2>&1 | grep "rpprof waves" | tail -1)
  echo "share $s: $t | $p" | tee -a gpurun_out/rpshare/log.txt
done

#!/bin/bash
# the wide kernel: A/B builds under gsn_amd/lib/variants and grid sizes, layer time + phase profile
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/w
: > gpurun_out/w/var.log
for so in gsn_amd/lib/libgsn_hip.so gsn_amd/lib/variants/libgsn_hip_*.so; do
  for grid in 256; do
    r=$(GSN_FUSED_GRID=$grid GSN_LIB_PATH=$so timeout 120 python scripts/bench_layer.py --wide --graphs 65536 --steps 10 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fused']['kernels_ms'], d['max_diff_over_max'])" 2>&1)
    p=$(GSN_FUSED_PROF=1 GSN_FUSED_GRID=$grid GSN_LIB_PATH=$so timeout 120 python scripts/bench_layer.py --wide --graphs 65536 --steps 8 2>&1 | grep "wprof range mid" | tail -1)
    echo "$(basename $so .so) grid $grid: $r | $p" | tee -a gpurun_out/w/var.log
  done
done

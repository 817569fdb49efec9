#!/bin/bash
# layer_g.hip variants: correctness (g_check without --skip-checks: "ALL OK") + time + phase profile per library under gsn_amd/lib/variants
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/gvar
: > gpurun_out/gvar/var2.log
for so in gsn_amd/lib/libgsn_hip.so gsn_amd/lib/variants/libgsn_hip_g*.so; do
  ok=$(GSN_LIB_PATH=$so timeout 300 python scripts/gpu/g_check.py --time 2>&1 | grep -E "ALL OK|FAIL|layer_g \[" | tr '\n' ' ')
  p=$(GSN_FUSED_PROF=1 GSN_LIB_PATH=$so timeout 120 python scripts/gpu/g_check.py --time --skip-checks 2>&1 | grep "gprof wave 0:" | tail -1)
  echo "$(basename $so .so): $ok | $p" | tee -a gpurun_out/gvar/var2.log
done

#!/bin/bash
# layer_g.hip: A/B builds under gsn_amd/lib/variants (RR_VARIANT_SRC=layer_g scripts/rr_variant.sh NAME -D...), layer time + phase profile
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/gvar
: > gpurun_out/gvar/var.log
for so in gsn_amd/lib/libgsn_hip.so gsn_amd/lib/variants/libgsn_hip_g*.so; do
  r=$(GSN_LIB_PATH=$so timeout 120 python scripts/gpu/g_check.py --time --skip-checks 2>&1 | grep "layer_g" | tail -1)
  p=$(GSN_FUSED_PROF=1 GSN_LIB_PATH=$so timeout 120 python scripts/gpu/g_check.py --time --skip-checks 2>&1 | grep "gprof wave 0" | tail -1)
  echo "$(basename $so .so): $r | $p" | tee -a gpurun_out/gvar/var.log
done

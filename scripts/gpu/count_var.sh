#!/bin/bash
# counting kernel A/B builds (gsn_amd/lib/variants/libgsn_hip_c*.so against the product library): kernel time with the int64 rows + pack
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c
for rep in 1 2; do
for so in gsn_amd/lib/libgsn_hip.so gsn_amd/lib/variants/libgsn_hip_c*.so; do
  echo "$(basename $so .so): $(GSN_LIB_PATH=$so timeout 300 python scripts/gpu/count_ab.py 2>&1 | grep 'int64 rows True  pack True' | tr '\n' ' ')" | tee -a gpurun_out/c/count_var.log
done
done

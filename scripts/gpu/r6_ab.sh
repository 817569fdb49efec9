#!/bin/bash
# r06: A/B of two builds of the library on one box (scripts/gpu/r6_step.py): r6_ab.sh VARIANT [VARIANT ...]; "default" = gsn_amd/lib/libgsn_hip.so
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6ab
for round in 1 2; do
for v in "$@"; do
  if [ $v = default ]; then unset GSN_LIB_PATH; else export GSN_LIB_PATH=gsn_amd/lib/variants/libgsn_hip_$v.so; fi
  echo "== $v" | tee -a gpurun_out/r6ab/ab.log
  timeout 240 python scripts/gpu/r6_step.py 2>&1 | grep -E "equal|one-call step  |count \(ids|count \+ side" | tail -4 | tee -a gpurun_out/r6ab/ab.log
done
done

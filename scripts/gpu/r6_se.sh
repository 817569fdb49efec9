#!/bin/bash
# r06: counting workgroups per side workgroup (GSN_SIDE_EVERY builds: scripts/rr_variant.sh seN -DGSN_SIDE_EVERY=N with RR_VARIANT_SRC=count)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6f
for v in default se2 se3 se6 se8 default; do
  if [ $v = default ]; then unset GSN_LIB_PATH; else export GSN_LIB_PATH=gsn_amd/lib/variants/libgsn_hip_$v.so; fi
  echo "== $v" | tee -a gpurun_out/r6f/se.log
  timeout 240 python scripts/gpu/r6_step.py 2>&1 | grep -E "one-call step  |count \(ids|count \+ side" | tail -4 | tee -a gpurun_out/r6f/se.log
done

#!/bin/bash
# r06: workgroups per call (slab length) and the VALU-per-MFMA hint of wgrad_f16x3_kernel, one process per setting
cd "$(dirname "$0")/../.."
for wgs in 256 512 768 1024 1536 2048 4096; do
  echo "== GSN_WGRAD16_WGS=$wgs GSN_WGRAD_WGS=$wgs"
  GSN_WGRAD16_WGS=$wgs GSN_WGRAD_WGS=$wgs timeout 200 python scripts/gpu/r6_wgrad16.py 2>&1 | grep -E "M 105083" | grep -v pre-pass
done
for v in 2 4 6; do
  echo "== GSN_WGRAD16_VALU=$v"
  GSN_WGRAD16_VALU=$v timeout 200 python scripts/gpu/r6_wgrad16.py 2>&1 | grep -E "M 105083.*f16x3"
done

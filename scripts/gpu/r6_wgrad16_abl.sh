#!/bin/bash
# r06: ablations of wgrad_f16x3_kernel (GSN_WGRAD16_DBG bits: 1 no atomics, 2 no products, 4 no loads behind the prologue, 8 no staging)
cd "$(dirname "$0")/../.."
for d in 0 1 2 4 8 3 6 12 14 15; do
  echo "== GSN_WGRAD16_DBG=$d"
  GSN_WGRAD16_DBG=$d timeout 200 python scripts/gpu/r6_wgrad16.py 2>&1 | grep -E "M 105083 N 300.*f16x3"
done

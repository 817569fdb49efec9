#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5i
for m in 0 1; do GSN_LINEAR_F16X3_STATS=$m timeout 600 python scripts/gpu/diag_ogb300.py 2>&1 | tail -12; done > gpurun_out/r5i/diag.log
cat gpurun_out/r5i/diag.log

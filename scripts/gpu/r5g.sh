#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5g
(timeout 300 python scripts/gpu/stats_ab.py 2>&1 | tail -12) > gpurun_out/r5g/stats_ab.log
cat gpurun_out/r5g/stats_ab.log

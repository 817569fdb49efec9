#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5p
(timeout 300 python scripts/gpu/glue_full_model.py 2>&1 | grep -v amdgpu | tail -70) > gpurun_out/r5p/glue.log
cat gpurun_out/r5p/glue.log | cut -c1-260

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_gpu.py -x -q -k "wide" 2>&1 | tail -30 | tee gpurun_out/wtest.log

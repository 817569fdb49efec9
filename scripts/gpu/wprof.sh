#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/w
(GSN_FUSED_PROF=1 timeout 300 python scripts/bench_layer.py --wide --graphs 65536 --steps 16 2>&1 | grep wprof | tail -2) | tee gpurun_out/w/prof.log
timeout 300 python scripts/bench_layer.py --wide --graphs 65536 --steps 20 2>&1 | tail -1 | tee gpurun_out/w/bench_layer_wide_65536.json

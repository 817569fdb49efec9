#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/w
timeout 300 python scripts/bench_layer.py --wide --graphs 65536 --steps 20 2>&1 | tail -1 | tee gpurun_out/w/bench_layer_wide_65536.json
timeout 300 python scripts/bench_layer.py --wide --graphs 16384 --steps 20 2>&1 | tail -1 | tee gpurun_out/w/bench_layer_wide_16384.json
(GSN_FUSED_PROF=1 timeout 300 python scripts/bench_layer.py --wide --graphs 65536 --steps 16 2>&1 | grep wprof | tail -2) | tee gpurun_out/w/prof.log
timeout 300 python scripts/profile_full_model.py 2>&1 | tail -1 | tee gpurun_out/w/full_model.json

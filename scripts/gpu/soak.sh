#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/soak
GSN_CHAIN_TRACE=1 timeout 1500 python tests/soak_layers.py 1000 200 --wide > gpurun_out/soak/layers_wide.log 2> gpurun_out/soak/layers_wide.err
grep -c "layer_fused_kernel_w " gpurun_out/soak/layers_wide.err
tail -3 gpurun_out/soak/layers_wide.log
timeout 1500 python tests/soak_count.py 700 300 2>&1 | tail -3 | tee gpurun_out/soak/count.log
timeout 900 python tests/soak_layers.py 2000 150 2>&1 | tail -2 | tee gpurun_out/soak/layers.log

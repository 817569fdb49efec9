#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4soak2
timeout 1500 python tests/soak_grads.py 6100 2000 2>&1 | tail -2 | tee gpurun_out/r4soak2/grads.log
timeout 1500 python tests/soak_layers.py 6200 2000 2>&1 | tail -2 | tee gpurun_out/r4soak2/layers.log
timeout 900 python tests/soak_layers.py 6300 400 --wide 2>&1 | tail -2 | tee gpurun_out/r4soak2/layers_wide.log
timeout 1500 python tests/soak_count.py 6400 1000 2>&1 | tail -2 | tee gpurun_out/r4soak2/count.log
timeout 900 python scripts/soak_dense.py 6500 600 2>&1 | tail -2 | tee gpurun_out/r4soak2/dense.log

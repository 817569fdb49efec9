#!/bin/bash
# soaks with fresh seeds after the round-4 changes (propagate U4 / segment sums, embedding paths, wgrad slabs, BatchNorm adjoint from h, arenas)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4soak
timeout 900 python tests/soak_grads.py 4100 600 2>&1 | tail -2 | tee gpurun_out/r4soak/grads.log
timeout 900 python tests/soak_layers.py 4200 400 2>&1 | tail -2 | tee gpurun_out/r4soak/layers.log
timeout 600 python tests/soak_layers.py 4300 100 --wide 2>&1 | tail -2 | tee gpurun_out/r4soak/layers_wide.log
timeout 900 python tests/soak_count.py 4400 300 2>&1 | tail -2 | tee gpurun_out/r4soak/count.log
timeout 600 python scripts/soak_dense.py 4500 300 2>&1 | tail -2 | tee gpurun_out/r4soak/dense.log

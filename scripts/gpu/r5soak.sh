#!/bin/bash
# soaks with fresh seeds over the session's kernels (pipelined propagate, split-K input gradients, wgrad pipeline, grid-stride row pre-pass)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5soak
(timeout 900 python tests/soak_grads.py 91000 400 2>&1 | tail -3) | tee gpurun_out/r5soak/grads.log
(timeout 900 python tests/soak_layers.py 92000 400 2>&1 | tail -2) | tee gpurun_out/r5soak/layers.log
(timeout 600 python scripts/soak_dense.py 93000 60 2>&1 | tail -2) | tee gpurun_out/r5soak/dense.log
(timeout 600 python tests/soak_count.py 94000 150 2>&1 | tail -2) | tee gpurun_out/r5soak/count.log

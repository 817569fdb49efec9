#!/bin/bash
# soaks with fresh seeds over the session's kernels (pipelined propagate, split-K input gradients, wgrad pipeline, grid-stride row pre-pass,
# pack-only identifiers); arguments: first seed block (default 91), scale (default 1)
cd ${GRAFT_REPO_ROOT:-.}
S=${1:-91}; K=${2:-1}
mkdir -p gpurun_out/r5soak
(timeout 1500 python tests/soak_grads.py ${S}1000 $((400 * K)) 2>&1 | tail -3) | tee gpurun_out/r5soak/grads.log
(timeout 1500 python tests/soak_layers.py ${S}2000 $((400 * K)) 2>&1 | tail -2) | tee gpurun_out/r5soak/layers.log
(timeout 900 python scripts/soak_dense.py ${S}3000 $((60 * K)) 2>&1 | tail -2) | tee gpurun_out/r5soak/dense.log
(timeout 900 python tests/soak_count.py ${S}4000 $((150 * K)) 2>&1 | tail -2) | tee gpurun_out/r5soak/count.log
(timeout 900 python tests/soak_layers.py ${S}5000 $((100 * K)) --wide 2>&1 | tail -2) | tee gpurun_out/r5soak/layers_wide.log

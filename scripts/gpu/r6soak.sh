#!/bin/bash
# round 6: soaks with fresh seeds over the final library (side workgroups, plane paths of the dense stages, everything r5soak.sh covers);
# arguments: first seed block (default 61), scale (default 1)
cd ${GRAFT_REPO_ROOT:-.}
S=${1:-61}; K=${2:-1}
mkdir -p gpurun_out/r6soak
(timeout 1500 python tests/soak_grads.py ${S}1000 $((400 * K)) 2>&1 | tail -3) | tee gpurun_out/r6soak/grads.log
(timeout 1500 python tests/soak_layers.py ${S}2000 $((400 * K)) 2>&1 | tail -2) | tee gpurun_out/r6soak/layers.log
(timeout 900 python scripts/soak_dense.py ${S}3000 $((60 * K)) 2>&1 | tail -2) | tee gpurun_out/r6soak/dense.log
(timeout 900 python tests/soak_count.py ${S}4000 $((150 * K)) 2>&1 | tail -2) | tee gpurun_out/r6soak/count.log
(timeout 900 python tests/soak_layers.py ${S}5000 $((100 * K)) --wide 2>&1 | tail -2) | tee gpurun_out/r6soak/layers_wide.log
(timeout 900 python tests/soak_side.py ${S}6000 $((150 * K)) 2>&1 | tail -2) | tee gpurun_out/r6soak/side.log
(timeout 900 python tests/soak_planes.py ${S}7000 $((600 * K)) 2>&1 | tail -2) | tee gpurun_out/r6soak/planes.log
(GSN_WGRAD16_DMA=1 GSN_L16_WIDE=1 timeout 900 python tests/soak_planes.py ${S}8000 $((300 * K)) 2>&1 | tail -2) | tee gpurun_out/r6soak/planes_optin.log

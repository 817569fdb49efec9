#!/bin/bash
# round 5: layer_g workgroups take equal shares of nodes (64-ary search over node_ptr): check + timing at 65 536 and 16 384 graphs, A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5r
(timeout 600 python -m pytest tests/test_fused_gpu.py -q -m gpu --tb=short 2>&1 | tail -5) | tee gpurun_out/r5r/tests.log | cut -c1-300
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export GSN_G_SPLIT_GRAPHS=1; else unset GSN_G_SPLIT_GRAPHS; fi
  echo "BY_GRAPHS=$v"; timeout 300 python scripts/gpu/g_check.py --time 2>&1 | grep -E "layer_g \[|ALL OK|element-wise"
  timeout 300 python scripts/profile_full_model.py 2>&1 | tail -1 | cut -c1-120
done | tee gpurun_out/r5r/ab.log

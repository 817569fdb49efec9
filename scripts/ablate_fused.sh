#!/bin/bash
# Diagnostic: kernel time of the one-launch layer with parts of its work switched off (GSN_FUSED_ABLATE, diagnostic build only;
# the outputs are garbage).  1 E matrix+epilogue, 2 E split, 4 S0 staging, 8 S0 matrix, 16 per-node sums, 32 S1 matrix+stores,
# 64 S1 split, 128 E gathers+split.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for m in 0 1 2 4 8 16 32 64 128 3 12 96 41 86 255; do
  echo -n "ablate $m: "
  GSN_FUSED_PROF=1 GSN_FUSED_ABLATE=$m python scripts/bench_layer.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['fused']['kernels_ms'])"
done

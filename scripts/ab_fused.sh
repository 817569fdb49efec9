#!/bin/bash
# A/B of two builds of libgsn_hip.so on ONE box (boxes differ by a few per cent): gsn_amd/lib/ab/libA.so and libB.so are swapped
# in turn under scripts/bench_layer.py, three rounds each, interleaved.  Usage (through gpurun): bash scripts/ab_fused.sh
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
cp gsn_amd/lib/libgsn_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for v in A B; do
    cp gsn_amd/lib/ab/lib$v.so gsn_amd/lib/libgsn_hip.so
    echo -n "$v: "; python scripts/bench_layer.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['fused']['kernels_ms'], d['max_diff_over_max'])"
  done
done
cp /tmp/lib_keep.so gsn_amd/lib/libgsn_hip.so

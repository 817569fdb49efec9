#!/bin/bash
# round 4, first GPU call: pack16 parity tests, the fused tests (new split helpers), layer timings with / without the edge-epilogue pipeline
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4a
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_pack16_gpu.py -x -q > "$OUT/pack16_test.log" 2>&1 </dev/null
tail -15 "$OUT/pack16_test.log"
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_range_gpu.py -x -q > "$OUT/fused_test.log" 2>&1 </dev/null
tail -5 "$OUT/fused_test.log"
timeout 300 python scripts/bench_layer.py > "$OUT/layer.json" 2>"$OUT/layer.err" </dev/null
cat "$OUT/layer.json"
GSN_LIB_PATH=$ROOT/gsn_amd/lib/variants/libgsn_hip_nopipe.so timeout 300 python scripts/bench_layer.py > "$OUT/layer_nopipe.json" 2>"$OUT/layer_nopipe.err" </dev/null
cat "$OUT/layer_nopipe.json"
GSN_FUSED_PROF=1 timeout 300 python scripts/bench_layer.py --steps 16 > "$OUT/layer_prof.json" 2>"$OUT/layer_prof.err" </dev/null
grep -h "prof range" "$OUT/layer_prof.err" | tail -8

#!/bin/bash
# ER full-size property test, propagate rocprof passes (stats + FETCH/WRITE), full-model family timers and kernel stats
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3b
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_count_gpu.py -x -q -k "full_size" > "$OUT/er_test.log" 2>&1 </dev/null
tail -3 "$OUT/er_test.log"
timeout 300 python scripts/bench_propagate.py > "$OUT/propagate.json" 2>&1 </dev/null
timeout 300 python scripts/profile_full_model.py > "$OUT/full_model.json" 2>&1 </dev/null
tail -1 "$OUT/full_model.json"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o prop -- python $ROOT/scripts/bench_propagate.py > "$OUT/prop_r.log" 2>&1 </dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o propf -- python $ROOT/scripts/bench_propagate.py > "$OUT/prop_f.log" 2>&1 </dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o propw -- python $ROOT/scripts/bench_propagate.py > "$OUT/prop_w.log" 2>&1 </dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o fm -- python $ROOT/scripts/profile_full_model.py > "$OUT/fm_r.log" 2>&1 </dev/null
find "$OUT" -name "*stats*.csv" | head
cat "$OUT/propagate.json"

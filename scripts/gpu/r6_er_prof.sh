#!/bin/bash
# r06: BASELINE config 5 (ER G(128,1000) x 21 five-vertex patterns, 2 048 graphs per launch): phase profile (COUNT_PROF build:
# RR_VARIANT_SRC=count scripts/rr_variant.sh cprof -DCOUNT_PROF -DCOUNT_PROF_STEP), kernel stats and SQ counters of the launch
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6er
mkdir -p "$OUT"
cd "$ROOT"
CMD="python $ROOT/scripts/bench_counting_er.py --graphs 2048 --steps 5"
timeout 300 $CMD 2>/dev/null | tail -1 | tee "$OUT/line.json"
(GSN_LIB_PATH=gsn_amd/lib/variants/libgsn_hip_cprof.so timeout 600 python scripts/bench_counting_er.py --graphs 2048 --steps 17 2>&1 | grep countprof | tail -2) | tee "$OUT/phase.log"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o r -- $CMD > "$OUT/r.log" 2>&1 </dev/null
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU \
    --kernel-trace --output-format csv -d "$OUT" -o p -- $CMD > "$OUT/p.log" 2>&1 </dev/null
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU \
    --kernel-trace --output-format csv -d "$OUT" -o q -- $CMD > "$OUT/q.log" 2>&1 </dev/null
cd "$ROOT" && python scripts/summarise_profile.py "$OUT" "$OUT/er128" | tail -1
cat "$OUT/er128_pmc.csv"; grep count_kernel "$OUT/er128_kernel_stats.csv" | head -3

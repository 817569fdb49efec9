#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run through gpurun from the repo root):
#   r: --kernel-trace --stats   per-kernel durations
#   p: SQ counters              (own pass; PMC never combined with sys/hip/hsa traces)
#   f, w: FETCH_SIZE / WRITE_SIZE (own passes, as MI355X_MICROARCH.md's HBM section prescribes)
# Results land in gpurun_out/<tag>/ and are condensed by scripts/summarise_profile.py into profiles/.
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
SHORT="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o r -- $BENCH > "$OUT/r.log" 2>&1 </dev/null
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS \
    --kernel-trace --output-format csv -d "$OUT" -o p -- $SHORT > "$OUT/p.log" 2>&1 </dev/null
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU \
    --kernel-trace --output-format csv -d "$OUT" -o q -- $SHORT > "$OUT/q.log" 2>&1 </dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o f -- $SHORT > "$OUT/f.log" 2>&1 </dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o w -- $SHORT > "$OUT/w.log" 2>&1 </dev/null
cd "$ROOT" && timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" </dev/null
find "$OUT" -name "*.csv" | head -20
tail -1 "$OUT/bench.json"

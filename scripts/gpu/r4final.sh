#!/bin/bash
# last session of round 4: whole GPU suite + smoke, kernel stats of the replayed small-batch steps and of the config-4 step,
# rocprofv3 passes over bench.py (kernel stats, SQ counters, FETCH / WRITE), the bench line
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
mkdir -p gpurun_out/r4final
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r4final/all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r4final/smoke.log
cat gpurun_out/r4final/all.log gpurun_out/r4final/smoke.log
bash scripts/gpu/r4h.sh > gpurun_out/r4final/r4h.log 2>&1
grep -E "kernels [0-9]+ calls" gpurun_out/r4final/r4h.log
bash scripts/gpu/molhiv_prof.sh > gpurun_out/r4final/molhiv.log 2>&1
head -3 gpurun_out/r4final/molhiv.log | cut -c1-250
bash scripts/profile_bench.sh r4prof > gpurun_out/r4final/prof.log 2>&1
tail -1 gpurun_out/r4prof/bench.json | cut -c1-400

#!/bin/bash
# round 5, last validation: whole GPU suite + smoke, rocprofv3 passes over bench.py + the bench line, config-4 kernel stats
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
mkdir -p gpurun_out/r5final2
(timeout 1800 python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -4) > gpurun_out/r5final2/all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r5final2/smoke.log
cat gpurun_out/r5final2/all.log gpurun_out/r5final2/smoke.log | cut -c1-400
bash scripts/profile_bench.sh r5prof > gpurun_out/r5final2/prof.log 2>&1
tail -1 gpurun_out/r5prof/bench.json | cut -c1-300
bash scripts/gpu/molhiv_prof.sh > gpurun_out/r5final2/molhiv.log 2>&1
head -3 gpurun_out/r5final2/molhiv.log | cut -c1-250

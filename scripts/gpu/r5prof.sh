#!/bin/bash
# round 5: rocprofv3 passes over bench.py (kernel stats, SQ counters, FETCH / WRITE) + the bench line, then layer_g timing
cd ${GRAFT_REPO_ROOT:-.}
bash scripts/profile_bench.sh r5prof > gpurun_out/r5prof.log 2>&1
tail -1 gpurun_out/r5prof/bench.json | cut -c1-3000
(timeout 300 python scripts/gpu/g_check.py --time 2>&1 | tail -15) > gpurun_out/r5prof/gcheck.log
cat gpurun_out/r5prof/gcheck.log

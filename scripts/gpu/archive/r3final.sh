#!/bin/bash
# whole GPU suite, smoke(), rocprofv3 passes over bench.py (stats, SQ counters, FETCH / WRITE) and the bench line; full-model kernel stats
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3t
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r3t/all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/r3t/smoke.log
bash scripts/profile_bench.sh r3prof > gpurun_out/r3t/profile.log 2>&1
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r3prof -o fm -- python $ROOT/scripts/profile_full_model.py > $ROOT/gpurun_out/r3prof/fm.log 2>&1 </dev/null
cd $ROOT
cat gpurun_out/r3t/all.log gpurun_out/r3t/smoke.log; tail -3 gpurun_out/r3t/profile.log | cut -c1-600; tail -1 gpurun_out/r3prof/fm.log

#!/bin/bash
# round 5: the whole GPU suite (no -x, one line per failure) + smoke + a short bench
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5k
(timeout 2400 python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -40) > gpurun_out/r5k/all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r5k/smoke.log
cat gpurun_out/r5k/all.log gpurun_out/r5k/smoke.log | cut -c1-700
(timeout 900 python bench.py 2>gpurun_out/r5k/bench.err | tail -1) > gpurun_out/r5k/bench.json
cut -c1-300 gpurun_out/r5k/bench.json; tail -3 gpurun_out/r5k/bench.err

#!/bin/bash
# the whole GPU suite, smoke(), the bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3t
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r3t/all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/r3t/smoke.log
timeout 900 python bench.py > gpurun_out/r3t/bench.json 2> gpurun_out/r3t/bench.err
cat gpurun_out/r3t/all.log gpurun_out/r3t/smoke.log; tail -c 3000 gpurun_out/r3t/bench.json

#!/bin/bash
# whole GPU suite + smoke + the bench line (round 4 checkpoints)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4t
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r4t/all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/r4t/smoke.log
(timeout 900 python bench.py 2> gpurun_out/r4t/bench.err | tail -1) > gpurun_out/r4t/bench.json
cat gpurun_out/r4t/all.log gpurun_out/r4t/smoke.log; cut -c1-1500 gpurun_out/r4t/bench.json

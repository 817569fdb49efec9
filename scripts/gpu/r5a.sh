#!/bin/bash
# round 5 baseline: bench line, layer_g check + timing, then the whole GPU suite + smoke
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5a
(timeout 600 python bench.py 2>&1 | tail -1) > gpurun_out/r5a/bench.json
cut -c1-600 gpurun_out/r5a/bench.json
(timeout 300 python scripts/gpu/g_check.py --time 2>&1 | tail -15) > gpurun_out/r5a/gcheck.log
cat gpurun_out/r5a/gcheck.log
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6) > gpurun_out/r5a/all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r5a/smoke.log
cat gpurun_out/r5a/all.log gpurun_out/r5a/smoke.log

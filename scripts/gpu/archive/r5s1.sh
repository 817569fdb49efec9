#!/bin/bash
# session 5 of round 5: ATen glue call sites of the small-batch steps + relu-sum propagate A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s1
(timeout 300 python scripts/gpu/glue_trace.py molhiv 2>&1 | tail -80) > gpurun_out/r5s1/glue_molhiv.txt
(timeout 300 python scripts/gpu/glue_trace.py zinc 2>&1 | tail -80) > gpurun_out/r5s1/glue_zinc.txt
(timeout 600 python scripts/gpu/prop_rs.py 2>&1 | tail -30) > gpurun_out/r5s1/prop_rs.txt
cat gpurun_out/r5s1/prop_rs.txt

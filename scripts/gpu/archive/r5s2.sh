#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s2
(timeout 900 python scripts/gpu/prop_cp.py 2>&1 | grep -v amdgpu.ids | tail -60) > gpurun_out/r5s2/prop_cp.txt
cat gpurun_out/r5s2/prop_cp.txt
(timeout 300 python scripts/gpu/prop_rs.py 0 16,2,0 16,1,0 0 2>&1 | tail -5) > gpurun_out/r5s2/prop_rs.txt
cat gpurun_out/r5s2/prop_rs.txt
(timeout 900 python -m pytest tests -q -m gpu -x --tb=short -k "propag or csr or pool or readout or gin or ogb" 2>&1 | tail -8) > gpurun_out/r5s2/tests.txt
cat gpurun_out/r5s2/tests.txt

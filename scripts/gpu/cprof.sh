#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c
(GSN_LIB_PATH=gsn_amd/lib/variants/libgsn_hip_cprof.so timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | grep countprof | tail -3) | tee gpurun_out/c/countprof.log
(GSN_COUNT_PAIR=0 GSN_LIB_PATH=gsn_amd/lib/variants/libgsn_hip_cprof.so timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | grep countprof | tail -3) | tee -a gpurun_out/c/countprof.log

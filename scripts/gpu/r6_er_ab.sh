#!/bin/bash
# r06: BASELINE config 5 (ER G(128,1000) x 21 five-vertex patterns) with two builds of the library on one box: r6_er_ab.sh VARIANT ...
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6er
for round in 1 2; do
for v in "$@"; do
  if [ $v = default ]; then unset GSN_LIB_PATH; else export GSN_LIB_PATH=gsn_amd/lib/variants/libgsn_hip_$v.so; fi
  for mode in vertex edge; do
    echo -n "$v $mode: " | tee -a gpurun_out/r6er/ab.log
    timeout 300 python scripts/bench_counting_er.py --graphs 2048 --steps 5 --mode $mode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['graphs_per_s'], d['ms_per_launch'])" | tee -a gpurun_out/r6er/ab.log
  done
done
done

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s7


{
for s in 0 1 0 1; do
  for w in molhiv zinc; do
    b=128; [ $w = molhiv ] && b=32
    echo "== $w SPLITK=$s"
    GSN_LINEAR_SPLITK=$s timeout 300 python scripts/train_step_$w.py --batch $b --steps 300 --warmup 3 --graph 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  done
done
} > gpurun_out/r5s7/replay.txt 2>&1
cat gpurun_out/r5s7/replay.txt

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s8
{
for r in 2 3 4 6 8 12; do
  echo "== molhiv RANGES=$r"
  GSN_LINEAR_SPLITK_RANGES=$r timeout 300 python scripts/train_step_molhiv.py --batch 32 --steps 300 --warmup 3 --graph 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
for r in 2 3 4; do
  echo "== zinc RANGES=$r"
  GSN_LINEAR_SPLITK_RANGES=$r timeout 300 python scripts/train_step_zinc.py --batch 128 --steps 300 --warmup 3 --graph 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
} > gpurun_out/r5s8/ranges.txt 2>&1
cat gpurun_out/r5s8/ranges.txt

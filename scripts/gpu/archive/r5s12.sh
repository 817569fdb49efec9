#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s12
{
for f in 0 4 5 6 0 4 5 6; do
  echo "== WGRAD_PIPE=$f"
  GSN_WGRAD_PIPE=$f timeout 300 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
for w in 1024 1536 3072 4096; do
  echo "== WGRAD_PIPE=5 WGS=$w"
  GSN_WGRAD_WGS=$w timeout 300 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
} > gpurun_out/r5s12/ab.txt 2>&1
cat gpurun_out/r5s12/ab.txt

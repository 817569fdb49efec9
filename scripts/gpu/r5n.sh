#!/bin/bash
# round 5: whole suite after the small launch-side changes (status words, one-hot groups) and the 16-byte-store embedding forward; config-4 step A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5n
(timeout 1800 python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -6) | tee gpurun_out/r5n/all.log | cut -c1-400
for v in 1 0; do
  if [ $v = 1 ]; then export GSN_EMBED_NOVEC4=1; else unset GSN_EMBED_NOVEC4; fi
  echo "NOVEC4=$v"; timeout 600 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>/dev/null | tail -1 | cut -c120-200
done | tee gpurun_out/r5n/steps.log
unset GSN_EMBED_NOVEC4
bash scripts/gpu/molhiv_prof.sh > gpurun_out/r5n/molhiv.log 2>&1
grep -E "embed|^\{" gpurun_out/r5n/molhiv.log | cut -c1-200

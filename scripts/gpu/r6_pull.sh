#!/bin/bash
# r06: GSN_PULL_BATCH (idle lanes that wait before the pool's pull arm runs) after the core filter, on the counting launch of the step
cd ${GRAFT_REPO_ROOT:-.}
for r in 1 2; do for p in 8 1 2 4 12 16 24; do
  echo -n "GSN_PULL_BATCH=$p: "; GSN_PULL_BATCH=$p timeout 240 python scripts/gpu/r6_step.py 2>&1 | grep -E "^count \+ side" | tail -1 | cut -c1-52
done; done

#!/bin/bash
# r06: launch parameters of the packed-row layer kernel on the one-call step (scripts/gpu/r6_step.py), one process per setting
cd ${GRAFT_REPO_ROOT:-.}
for s in 0.5 0.56 0.6 0.64 0.68; do
  echo -n "GSN_RP_OLD_SHARE=$s: "; GSN_RP_OLD_SHARE=$s timeout 240 python scripts/gpu/r6_step.py 2>&1 | grep -E "^one-call step  " | tail -1
done
for g in 128 512; do
  echo -n "GSN_FUSED_GRID=$g: "; GSN_FUSED_GRID=$g timeout 240 python scripts/gpu/r6_step.py 2>&1 | grep -E "^one-call step  " | tail -1
done

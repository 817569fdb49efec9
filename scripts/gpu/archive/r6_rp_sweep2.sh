cd ${GRAFT_REPO_ROOT:-.}
for r in 1 2; do for s in 0.6 0.52 0.54 0.56 0.58; do
  echo -n "GSN_RP_OLD_SHARE=$s: "; GSN_RP_OLD_SHARE=$s timeout 240 python scripts/gpu/r6_step.py 2>&1 | grep -E "^one-call step  " | tail -1 | cut -c1-50
done; done

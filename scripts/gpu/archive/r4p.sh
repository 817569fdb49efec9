#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for v in 0 8 16 32 0 16; do
  GSN_PROP_LPR=$v timeout 300 python scripts/bench_propagate.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('LPR=$v', [(c['case'][:22], c['ms']) for c in d])"
done

#!/bin/bash
# the replay == eager tests (ELU models, one attempt each since r06), many times over: failure messages only
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq 1 ${1:-20}); do
  timeout 300 python -m pytest tests/test_graphed_train_gpu.py -x -q -k "refilled or replay_equals_eager" --tb=line 2>&1 | grep -E "drifts|passed|failed|Error|loss" | cut -c1-400
done

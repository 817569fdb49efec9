#!/bin/bash
# the refilled-static-batch replay test, many times over: failure messages only
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq 1 ${1:-30}); do
  timeout 120 python -m pytest tests/test_graphed_train_gpu.py -x -q -k "refilled" --tb=line 2>&1 | grep -E "drifts|passed|failed|Error" | cut -c1-400
done

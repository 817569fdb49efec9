#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for m in test_big_batch_gpu test_chain_fuzz_gpu test_codes_gpu test_count_gpu test_dataset_gpu test_directed_gpu test_encoding_gpu test_end_to_end_gpu test_eval_grad_gpu test_fused_gpu; do
  r=$(timeout 900 python -m pytest tests/$m.py tests/test_graphed_train_gpu.py -x -q -k "not soak" 2>&1 | tail -1)
  echo "$m -> $r"
done

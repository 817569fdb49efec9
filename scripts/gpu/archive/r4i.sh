#!/bin/bash
# flat embedding backward: tests, replayed steps with and without it
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4i
timeout 900 python -m pytest tests/test_encoding_gpu.py tests/test_graphed_train_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r4i/test.log
cat gpurun_out/r4i/test.log
for f in 1 0; do
  echo "GSN_EMBED_BWD_FLAT=$f"
  GSN_EMBED_BWD_FLAT=$f timeout 300 python scripts/train_step_molhiv.py --batch 32 --steps 100 --warmup 5 --graph 2>&1 | tail -1 | cut -c1-230
  GSN_EMBED_BWD_FLAT=$f timeout 300 python scripts/train_step_molhiv.py --batch 32 --steps 50 --warmup 5 2>&1 | tail -1 | cut -c1-230
  GSN_EMBED_BWD_FLAT=$f timeout 300 python scripts/train_step_molhiv.py --batch 4096 --steps 20 --warmup 10 2>&1 | tail -1 | cut -c1-230
done | tee gpurun_out/r4i/steps.log

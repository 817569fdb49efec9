#!/bin/bash
# round 5: new tests, soaks with fresh seeds after the round-5 kernel changes (16-bit staged counts, train-mode fp16x3 stages, layer_g, the module split),
# then the out-of-bounds detector (caching allocator off) over the suite, bench.py and smoke()
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5soak
(timeout 900 python -m pytest tests/test_count_gpu.py tests/test_big_batch_gpu.py tests/test_model_gpu.py tests/test_graphed_train_gpu.py -q -m gpu --tb=line 2>&1 | tail -6) | tee gpurun_out/r5soak/tests.log | cut -c1-400
timeout 900 python tests/soak_grads.py 5100 600 2>&1 | tail -2 | tee gpurun_out/r5soak/grads.log
timeout 900 python tests/soak_layers.py 5200 400 2>&1 | tail -2 | tee gpurun_out/r5soak/layers.log
timeout 600 python tests/soak_layers.py 5300 150 --wide 2>&1 | tail -2 | tee gpurun_out/r5soak/layers_wide.log
timeout 900 python tests/soak_count.py 5400 400 2>&1 | tail -2 | tee gpurun_out/r5soak/count.log
timeout 600 python scripts/soak_dense.py 5500 300 2>&1 | tail -2 | tee gpurun_out/r5soak/dense.log
bash scripts/oob_check.sh > gpurun_out/r5soak/oob.log 2>&1; echo "oob rc $?" >> gpurun_out/r5soak/oob.log
tail -40 gpurun_out/r5soak/oob.log

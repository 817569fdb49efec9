#!/bin/bash
# propagate / eval-gradient tests, then the rocprofv3 passes over bench.py + the bench line + the full-model and training-step kernel stats
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3t
(timeout 1200 python -m pytest tests/test_layers_gpu.py tests/test_eval_grad_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3) > gpurun_out/r3t/prop.log
bash scripts/profile_bench.sh r3prof > gpurun_out/r3t/profile.log 2>&1
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r3prof -o fm -- python $ROOT/scripts/profile_full_model.py > $ROOT/gpurun_out/r3prof/fm.log 2>&1 </dev/null
cd $ROOT
bash scripts/gpu/molhiv_prof.sh > gpurun_out/r3t/molhiv.log 2>&1
cat gpurun_out/r3t/prop.log; tail -3 gpurun_out/r3t/profile.log | cut -c1-300; tail -1 gpurun_out/r3prof/fm.log; head -1 gpurun_out/r3t/molhiv.log | cut -c1-300

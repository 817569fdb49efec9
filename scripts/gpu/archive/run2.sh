mkdir -p gpurun_out/r3c
(timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_range_gpu.py -x -q -m gpu -k "fused" 2>&1 | tail -5) > gpurun_out/r3c/tests.log
(GSN_FUSED_PROF=1 timeout 300 python scripts/bench_layer.py --graphs 65536 --steps 16 2>&1 | grep -v amdgpu.ids | tail -4) > gpurun_out/r3c/prof.log
(timeout 300 python scripts/bench_layer.py --graphs 65536 2>&1 | grep -v amdgpu.ids | tail -2) > gpurun_out/r3c/base.log
(timeout 300 python scripts/bench_layer.py --graphs 65536 --float-inputs 2>&1 | grep -v amdgpu.ids | tail -2) > gpurun_out/r3c/float.log
for f in gpurun_out/r3c/*.log; do echo "== $f"; cat $f | cut -c1-400; done

mkdir -p gpurun_out/r3e
(timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_range_gpu.py -x -q -m gpu -k "fused" 2>&1 | tail -5) > gpurun_out/r3e/tests.log
(timeout 300 python scripts/bench_layer.py --graphs 65536 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-330) > gpurun_out/r3e/base.log
(timeout 300 python scripts/bench_layer.py --graphs 65536 --float-inputs 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-330) > gpurun_out/r3e/float.log
bash scripts/gpu/pmc_layer.sh r3e_pmc > gpurun_out/r3e/pmc.log 2>&1
for f in gpurun_out/r3e/*.log; do echo "== $f"; cat $f; done

mkdir -p gpurun_out/r3b
(GSN_FUSED_PROF=1 timeout 300 python scripts/bench_layer.py --graphs 65536 --steps 16 2>&1 | grep -v amdgpu.ids | tail -6) > gpurun_out/r3b/prof.log
for v in nosb pd8; do (GSN_LIB_PATH=gsn_amd/lib/variants/libgsn_hip_$v.so timeout 300 python scripts/bench_layer.py --graphs 65536 2>&1 | grep -v amdgpu.ids | tail -2) > gpurun_out/r3b/$v.log; done
(timeout 300 python scripts/bench_layer.py --graphs 65536 2>&1 | grep -v amdgpu.ids | tail -2) > gpurun_out/r3b/base.log
for f in gpurun_out/r3b/*.log; do echo "== $f"; cat $f | cut -c1-600; done

# time scripts/bench_layer.py with every A/B build under gsn_amd/lib/variants (and the product library)
mkdir -p gpurun_out/var
for so in gsn_amd/lib/variants/libgsn_hip_*.so gsn_amd/lib/libgsn_hip.so; do
  v=$(basename $so .so)
  r=$(GSN_LIB_PATH=$so timeout 120 python scripts/bench_layer.py --graphs 65536 $1 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fused']['kernels_ms'], d['max_diff_over_max'])" 2>&1)
  echo "$v: $r" | tee -a gpurun_out/var/times.log
done

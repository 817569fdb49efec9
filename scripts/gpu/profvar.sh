mkdir -p gpurun_out/prof
for so in gsn_amd/lib/variants/libgsn_hip_*.so gsn_amd/lib/libgsn_hip.so; do
  echo "== $(basename $so .so)" | tee -a gpurun_out/prof/profvar.log
  (GSN_LIB_PATH=$so GSN_FUSED_PROF=1 timeout 300 python scripts/bench_layer.py --graphs 65536 --steps 16 2>&1 | grep "rrprof range mid" | tail -2 | cut -c1-330) | tee -a gpurun_out/prof/profvar.log
done

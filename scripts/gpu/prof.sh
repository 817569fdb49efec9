mkdir -p gpurun_out/prof
(GSN_FUSED_PROF=1 timeout 300 python scripts/bench_layer.py --graphs 65536 --steps 16 $1 2>&1 | grep rrprof | tail -4) | tee gpurun_out/prof/prof.log

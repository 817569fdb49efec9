mkdir -p gpurun_out/r3t
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r3t/all.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/r3t/smoke.log
cat gpurun_out/r3t/all.log gpurun_out/r3t/smoke.log

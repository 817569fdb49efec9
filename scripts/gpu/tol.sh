mkdir -p gpurun_out/tol
(timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_layers_gpu.py -q -m gpu -k "train or backward or model" 2>&1 | tail -40) > gpurun_out/tol/model_layers.log
(timeout 1200 python -m pytest tests/test_big_batch_gpu.py -q -m gpu -k "300 or backward" 2>&1 | tail -30) > gpurun_out/tol/big.log
cat gpurun_out/tol/*.log

#!/bin/bash
# nt output stores in layer_rr: fused-layer tests, the bench line's layer figures
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4l
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_range_gpu.py tests/test_pack16_gpu.py -x -q 2>&1 | tail -4
(timeout 900 python bench.py 2> gpurun_out/r4l/bench.err | tail -1) > gpurun_out/r4l/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4l/bench.json")); k = d["kernels"]
print("value", d["value"], "ms", d["ms_per_step"], "layer", d["roofline"]["avg_launch_ms"])
for key in ("layer_alone_ms", "layer_float_inputs", "full_model_step", "layer_wide_d128", "fused_encoder_step"):
    print(key, json.dumps(k.get(key))[:200])
print(d["checked"])
PY

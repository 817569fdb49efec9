#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4r
timeout 1200 python -m pytest tests/test_rccl_gpu.py tests/test_graphed_train_gpu.py -x -q 2>&1 | tail -30 | cut -c1-400
(timeout 1200 python bench.py 2> gpurun_out/r4r/bench.err | tail -1) > gpurun_out/r4r/bench.json
tail -5 gpurun_out/r4r/bench.err | cut -c1-300
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4r/bench.json"))
k = d["kernels"]
print("value", d["value"], "ms", d["ms_per_step"], "roof", {x: d["roofline"].get(x) for x in ("kernel", "frac", "hbm_frac", "avg_launch_ms", "mfma_frac")})
for key in ("hip_graph_ms_per_step", "hip_graph_forked_ms_per_step", "step_with_int64_ids", "layer_alone_ms", "fused_encoder_step", "train_small_batch", "train_step_config4", "full_model_step", "layer_wide_d128", "layer_float_inputs", "hip_graph_note"):
    print(key, k.get(key))
print("checked", d["checked"])
PY

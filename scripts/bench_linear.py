#!/usr/bin/env python3
"""gsn_linear_fwd_hip alone (the any-shape dense stage): fp32-equivalent TFLOP/s at the shapes of the d = 300 ogb layers and
of the K = 260 message stage of the ZINC model's later layers.  Direct rows run the fp16x3 kernel (gsn_linear_f16x3_fwd_hip);
GSN_LINEAR_F16X3=0 selects the bf16x6 kernel, with GSN_LINEAR_BF16X6=0 the fp32-MFMA kernel."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsn_amd import flags, layers  # noqa: E402


def main():
    dev = torch.device("cuda")
    out = []
    shapes = ((196608, 300, 600), (196608, 600, 300), (1 << 20, 260, 128), (1 << 20, 128, 128), (380000, 128, 256))
    if os.environ.get("GSN_L16_SHAPES"):                      # (scripts/micro/l16_ablate.sh: one shape per run; or M:K:N)
        shapes = [tuple(int(v) for v in i.split(":")) if ":" in i else shapes[int(i)] for i in os.environ["GSN_L16_SHAPES"].split(",")]
    for M, K, N in shapes:
        x = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        st = layers._Stage(W, b, None, "relu", [(x, None)])
        f = lambda: layers._launch_stages([st], M)
        for _ in range(3):
            y = f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            y = f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        ref = torch.relu(x[:4096].double() @ W.double().T + b.double())
        err = float((y[:4096].double() - ref).abs().max() / ref.abs().max())
        out.append({"M": M, "K": K, "N": N, "ms": round(dt * 1e3, 3), "fp32_equivalent_TFLOPs": round(2.0 * M * K * N / dt / 1e12, 1),
                    "max_rel_err_vs_fp64": err})
    kern = "fp16x3" if flags.LINEAR_F16X3 else ("bf16x6" if os.environ.get("GSN_LINEAR_BF16X6", "1") != "0" else "fp32 mfma")
    print(json.dumps({"kernel": kern, "cases": out}))


if __name__ == "__main__":
    main()

"""r06: the fp16x3 product of the d = 300 ogb stages on 128 x 128 tiles (two workgroups per CU) and on 128 x 320 tiles (GSN_L16_WIDE, one per CU):
time of the product alone (rows split earlier) at 105 083 rows, 300 -> 600 and 600 -> 300."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsn_amd import _abi  # noqa: E402
from gsn_amd._dense import _f16x3_weights  # noqa: E402

L = _abi.lib()
dev = torch.device("cuda", 0)
for M, K, N in ((105083, 300, 600), (105083, 600, 300), (196608, 300, 600)):
    torch.manual_seed(0)
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    planes, col_inv = _f16x3_weights(W, W)
    y = torch.empty(M, N, device=dev)
    one = (_abi.gsn_block * 1)()
    one[0].data = x.data_ptr(); one[0].idx = None; one[0].idx32 = None; one[0].width = K
    sc = torch.empty(int(L.gsn_linear_f16x3_scratch_bytes(M, K)), dtype=torch.uint8, device=dev)
    _abi.check(L.gsn_linear_f16x3_split_rows_hip(M, 1, one, sc.data_ptr(), _abi.current_stream()), "split")
    stats = torch.zeros(2, N, dtype=torch.float64, device=dev)
    fns = {"product": lambda: _abi.check(L.gsn_linear_f16x3_fwd_presplit_hip(M, 1, one, planes.data_ptr(), col_inv.data_ptr(), b.data_ptr(), N, None, None, None, 1, sc.data_ptr(), y.data_ptr(), _abi.current_stream()), "p"),
           "product + statistics": lambda: _abi.check(L.gsn_linear_f16x3_fwd_stats_presplit_hip(M, 1, one, planes.data_ptr(), col_inv.data_ptr(), b.data_ptr(), N, sc.data_ptr(), y.data_ptr(), stats.data_ptr(), _abi.current_stream()), "s")}
    for name, fn in fns.items():
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("M %6d K %3d N %3d  %-22s %7.1f us  %6.1f TF/s fp32-equivalent" % (M, K, N, name, dt * 1e6, 2.0 * M * K * N / dt / 1e12), flush=True)
    ref = torch.relu(x[:4096].double() @ W.double().t() + b.double())
    fns["product"]()
    print("   max error over the row maximum: %.2e" % float(((y[:4096].double() - ref).abs() / ref.abs().amax(1, keepdim=True).clamp_min(1e-30)).max()))

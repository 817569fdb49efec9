"""r06: gsn_wgrad_f16x3_hip beside gsn_wgrad_hip at the d = 300 ogb shapes of BASELINE config 4 (105 083 rows, 300 <-> 600): time per call and
error against float64."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from gsn_amd import _abi  # noqa: E402
import test_wgrad16_gpu as T  # noqa: E402

L = _abi.lib()
SHAPES = ((105083, 300, 600),) if os.environ.get("R6_WGRAD16_ONE") else tuple((int(m), int(os.environ.get("R6_WGRAD16_N", "300")), int(os.environ.get("R6_WGRAD16_K", "600"))) for m in os.environ["R6_WGRAD16_M"].split(",")) if os.environ.get("R6_WGRAD16_M") else ((105083, 300, 600), (105083, 600, 300), (4096, 300, 600), (837, 600, 300))
for m, n_out, k in SHAPES:
    torch.manual_seed(0)
    gh = torch.randn(m, n_out, device="cuda") * 1e-4
    x = torch.relu(torch.randn(m, k, device="cuda"))
    sg, sx = T._split(gh), T._split(x)
    gw = torch.zeros(n_out, k, device="cuda")
    arr = (_abi.gsn_block * 1)()
    arr[0].data = x.data_ptr(); arr[0].idx = None; arr[0].idx32 = None; arr[0].width = k

    def new():
        _abi.check(L.gsn_wgrad_f16x3_hip(m, n_out, k, sg.data_ptr(), sx.data_ptr(), gw.data_ptr(), _abi.current_stream()), "f16")

    def old():
        _abi.check(L.gsn_wgrad_hip(m, n_out, gh.data_ptr(), 1, arr, gw.data_ptr(), _abi.current_stream()), "bf16")

    def split_both():
        T._split(gh); T._split(x)

    for name, fn in (("bf16x6", old), ("f16x3", new), ("the two pre-passes", split_both)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("M %6d N %3d K %3d  %-20s %8.1f us   %6.1f TF/s fp32-equivalent" % (m, n_out, k, name, dt * 1e6, 2.0 * m * n_out * k / dt / 1e12), flush=True)
    e16 = T._errs(T._wgrad16(gh, x), gh, x)[0]
    ebf = T._errs(T._wgrad_bf16(gh, x), gh, x)[0]
    ref = gh.double().t() @ x.double()
    e32 = (((gh.t() @ x).double() - ref).abs() / (gh.double().abs().t() @ x.double().abs())).max().item()
    print("   max error over sum|g||x|: f16x3 %.2e  bf16x6 %.2e  fp32 product %.2e" % (e16, ebf, e32), flush=True)

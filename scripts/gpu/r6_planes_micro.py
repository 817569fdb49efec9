"""r06: gsn_bn_act_planes_hip / gsn_bn_act_bwd_planes_hip alone at the d = 300 ogb shapes (105 083 rows, 600 / 300 columns) beside the passes they
replace (gsn_bn_act_hip + row pre-pass; gsn_bn_act_bwd_from_h_hip + row pre-pass): time per call and bytes moved."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsn_amd import _abi  # noqa: E402

L = _abi.lib()
st = _abi.current_stream
for m, c in ((105083, 600), (105083, 300)):
    torch.manual_seed(0)
    h = torch.randn(m, c, device="cuda"); g = torch.randn(m, c, device="cuda") * 1e-4
    mean, scale, shift, invstd = (torch.randn(c, device="cuda") * 0.1, torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.1, torch.rand(c, device="cuda") + 0.5)
    out = torch.empty_like(h); gh = torch.empty_like(h)
    scr = torch.empty(int(L.gsn_linear_f16x3_scratch_bytes(m, c)), dtype=torch.uint8, device="cuda")
    sums = torch.zeros(2, c, dtype=torch.float64, device="cuda"); gb = torch.zeros(c, dtype=torch.float64, device="cuda")
    one = (_abi.gsn_block * 1)()

    def split(t):
        one[0].data = t.data_ptr(); one[0].idx = None; one[0].idx32 = None; one[0].width = c
        _abi.check(L.gsn_linear_f16x3_split_rows_hip(m, 1, one, scr.data_ptr(), st()), "split")

    fns = {
        "fwd: bn_act (fp32 out)": lambda: _abi.check(L.gsn_bn_act_hip(m, c, h.data_ptr(), mean.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, out.data_ptr(), st()), "a"),
        "fwd: row pre-pass": lambda: split(out),
        "fwd: bn_act_planes": lambda: _abi.check(L.gsn_bn_act_planes_hip(m, c, h.data_ptr(), mean.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, None, scr.data_ptr(), st()), "b"),
        "bwd: reduce + apply (fp32 gH)": lambda: _abi.check(L.gsn_bn_act_bwd_from_h_hip(m, c, g.data_ptr(), h.data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, 1, sums.data_ptr(), gh.data_ptr(), gb.data_ptr(), st()), "c"),
        "bwd: row pre-pass": lambda: split(gh),
        "bwd: reduce + planes": lambda: _abi.check(L.gsn_bn_act_bwd_planes_hip(m, c, g.data_ptr(), h.data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, 1, sums.data_ptr(), scr.data_ptr(), gb.data_ptr(), st()), "d"),
    }
    for name, fn in fns.items():
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print("M %6d C %3d  %-32s %7.1f us" % (m, c, name, (time.perf_counter() - t0) / 20 * 1e6), flush=True)

"""r06: the weight gradient of a `general` layer's edge stage at config 2's training shape (194 284 edge rows, n_out 128, K = 128 + 128 + 12 + 4): blocks
gathered through edge_index (wgrad_bf16_kernel<true>) against the same rows assembled first (wgrad_bf16_pipe_kernel on plain rows)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsn_amd import _abi  # noqa: E402

L = _abi.lib()
dev = torch.device("cuda", 0)
N, E, d = 94832, 194284, 128
torch.manual_seed(0)
x = torch.randn(N, d, device=dev)
ids = torch.randn(E, 12, device=dev); ef = torch.randn(E, 4, device=dev)
ei = torch.randint(0, N, (2, E), device=dev)
ei = ei[:, torch.argsort(ei[1])].contiguous()
gh = torch.randn(E, 128, device=dev) * 1e-3
gw = torch.zeros(128, 272, device=dev)


def blocks(gathered):
    arr = (_abi.gsn_block * 4)()
    keep = []
    if gathered:
        srcs = [(x, ei[1].contiguous()), (x, ei[0].contiguous()), (ids, None), (ef, None)]
    else:
        srcs = [(x[ei[1]].contiguous(), None), (x[ei[0]].contiguous(), None), (ids, None), (ef, None)]
    for i, (t, ix) in enumerate(srcs):
        keep += [t, ix]
        arr[i].data = t.data_ptr(); arr[i].idx = ix.data_ptr() if ix is not None else None; arr[i].idx32 = None; arr[i].width = t.shape[1]
    return arr, keep


for name, g in (("gathered blocks", True), ("assembled rows", False)):
    arr, keep = blocks(g)
    f = lambda: _abi.check(L.gsn_wgrad_hip(E, 128, gh.data_ptr(), 4, arr, gw.data_ptr(), _abi.current_stream()), "wgrad")
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%-16s %7.1f us  %6.1f TF/s fp32-equivalent" % (name, dt * 1e6, 2.0 * E * 128 * 272 / dt / 1e12), flush=True)

import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_wgrad16_gpu as T
from gsn_amd import _abi
torch.manual_seed(0)
m, n, k = 32, 64, 64
bad = []
for r in range(m):
    gh = torch.zeros(m, n, device="cuda"); x = torch.zeros(m, k, device="cuda")
    gh[r] = torch.arange(1, n + 1, device="cuda").float()
    x[r] = torch.arange(1, k + 1, device="cuda").float() * 0.5
    gw = T._wgrad16(gh, x)
    ref = gh.t() @ x
    if (gw - ref).abs().max().item() > 0: bad.append(r)
print("single rows failing:", bad)
# decode a scratch
def decode(t):
    mm, kk = t.shape
    s = T._split(t)
    m_pad = int(_abi.lib().gsn_linear_f16x3_mpad(mm))
    kpad = int(_abi.lib().gsn_linear_f16x3_kpad(kk))
    inv = s[:4 * m_pad].view(torch.float32)[:mm]
    pl = s[4 * m_pad:].view(torch.float16).view(m_pad, kpad // 32, 2, 32)[:mm].float()
    v = (pl[:, :, 0] + pl[:, :, 1]).reshape(mm, kpad)[:, :kk] * inv[:, None]
    return v
x = torch.randn(100, 72, device="cuda") * torch.logspace(-3, 3, 100, device="cuda")[:, None]
print("pre-pass decode max rel err", ((decode(x) - x).abs() / x.abs().clamp_min(1e-30)).max().item())
# pairs of rows
for a, b in ((0, 1), (0, 2), (0, 15), (1, 3), (0, 16)):
    gh = torch.zeros(m, n, device="cuda"); x = torch.zeros(m, k, device="cuda")
    gh[a] = 1; gh[b] = 2; x[a] = 3; x[b] = 5
    gw = T._wgrad16(gh, x)
    print("rows", a, b, "gw[0,0]", gw[0, 0].item(), "expect", 13.0)

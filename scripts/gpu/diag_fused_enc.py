"""Where do the gradients of the fused-edge-encoder path differ from the two-encoder path?  (dense = a bug; a few rows = ReLU kinks)"""
import importlib.util, os, sys, types
import torch
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, REPO)
spec = importlib.util.spec_from_file_location("m", os.path.join(REPO, "scripts", "train_step_molhiv.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from gsn_amd import flags
dev = torch.device("cuda", 0)
d, batch, train = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "train"
act = sys.argv[4] if len(sys.argv) > 4 else None
def run(fuse):
    flags.FUSE_EDGE_ENCODERS = fuse
    torch.manual_seed(7)
    model, data, params, opt, loss_of, N, E = m.build(types.SimpleNamespace(batch=batch, layers=3, d=d, optimizer="sgd"), dev, 0, dropout=0.0)
    model.train(train)
    out, xs = model(data, return_intermediate=True)
    loss = (out * torch.linspace(-1.0, 1.0, out.numel(), device=dev).view_as(out)).sum()
    loss.backward()
    return out.detach(), [x.detach() for x in xs], {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
o0, x0, g0 = run(False)
o1, x1, g1 = run(True)
print("out", (o1 - o0).abs().max().item(), o0.abs().max().item())
for i, (a, b) in enumerate(zip(x0, x1)):
    print("x_interm", i, (a - b).abs().max().item(), a.abs().max().item())
for n in g0:
    s = g0[n].abs().max().item(); df = (g1[n] - g0[n]).abs()
    bad = (df > 1e-5 * s).sum().item()
    if df.max().item() > 2e-5 * s:
        rows = (df.view(df.shape[0], -1).max(1).values > 1e-5 * s).sum().item() if df.dim() > 1 else -1
        print("%-60s max diff %.3e scale %.3e  elements off %d / %d  rows %d" % (n, df.max().item(), s, bad, df.numel(), rows))

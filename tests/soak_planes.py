#!/usr/bin/env python3
"""Soak run of the dense stage lists under autograd at row counts where the plane paths of r06 apply (test infrastructure, not collected by
pytest): random depth (1-3 stages), widths (multiples of 4 and not, 64 .. 700), BatchNorm per stage (none / train / eval with or without
affine gradients), smooth activations (elu / tanh / identity: two correct implementations cannot land on different sides of a kink), 1-3
input blocks, 9 000 .. 40 000 rows, inputs with or without gradients, row magnitudes over four decades -- every gradient against a float64
PyTorch evaluation of models_misc.py:41-59's stages, with the plane switches on and off (the off run bounds what fp32 kernels do).

    python tests/soak_planes.py [first_seed] [n_seeds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsn_amd import _abi, _autograd, flags  # noqa: E402
from gsn_amd._dense import _Stage  # noqa: E402

ACTS = {"elu": torch.nn.functional.elu, "tanh": torch.tanh, "identity": lambda t: t}


def one_case(seed):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    m = int(rng.integers(9000, 40000))
    n_stage = int(rng.integers(1, 4))
    widths = [int(rng.choice([int(rng.integers(16, 175)) * 4, int(rng.integers(64, 700))])) for _ in range(n_stage + 1)]
    if rng.random() < 0.5:
        widths[0] = int(rng.integers(16, 100)) * 4
    n_blocks = int(rng.integers(1, 4))
    cuts = sorted(set(int(c) // 4 * 4 for c in rng.integers(4, max(widths[0] - 3, 5), n_blocks - 1))) if widths[0] >= 16 else []
    cuts = [c for c in cuts if 0 < c < widths[0]]
    bw = [b - a for a, b in zip([0] + cuts, cuts + [widths[0]])]
    rows = torch.logspace(-2, 2, m, device="cuda")[torch.randperm(m, device="cuda")][:, None]
    want_x = bool(rng.random() < 0.7)
    blocks = [(torch.randn(m, w, device="cuda") * rows).requires_grad_(want_x) for w in bw]
    lins, bns, acts, modes = [], [], [], []
    for i in range(n_stage):
        lins.append(torch.nn.Linear(widths[i], widths[i + 1], bias=bool(rng.random() < 0.8)).cuda())
        mode = str(rng.choice(["none", "train", "eval", "eval_frozen"]))
        bn = None
        if mode != "none":
            bn = torch.nn.BatchNorm1d(widths[i + 1]).cuda()
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0)
            bn.train(mode == "train")
            if mode == "eval_frozen":
                bn.weight.requires_grad_(False); bn.bias.requires_grad_(False)
        bns.append(bn); modes.append(mode)
        acts.append(str(rng.choice(["elu", "tanh", "identity"])))
    wout = torch.randn(m, widths[-1], device="cuda")
    one_case.desc = "m=%d widths=%s blocks=%s bn=%s act=%s want_x=%d" % (m, widths, bw, modes, acts, want_x)
    params = [p for lin in lins for p in lin.parameters()] + [p for bn in bns if bn is not None for p in bn.parameters() if p.requires_grad]
    leaves = ([t for t in blocks] if want_x else []) + params

    def run():
        for t in leaves:
            t.grad = None
        stages = [_Stage(lins[0].weight, lins[0].bias, bns[0], acts[0], blocks=[(t, None) for t in blocks])]
        stages += [_Stage(lins[i].weight, lins[i].bias, bns[i], acts[i]) for i in range(1, n_stage)]
        snap = [None if bn is None else (bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()) for bn in bns]
        y = _autograd.run_stages_autograd(stages, m, True)
        (y * wout).sum().backward()
        for bn, sn in zip(bns, snap):               # (the second run must see the same running statistics)
            if bn is not None:
                bn.running_mean.copy_(sn[0]); bn.running_var.copy_(sn[1]); bn.num_batches_tracked.copy_(sn[2])
        return y.detach().clone(), [t.grad.clone() for t in leaves]

    called = []
    real = _abi.check
    _abi.check = lambda rc, what="": (called.append(what), real(rc, what))[1]
    try:
        flags.WGRAD_F16X3 = flags.BN_BWD_PLANES = flags.BN_ACT_PLANES = True
        y_new, g_new = run()
        used = sorted(set(c for c in called if "planes" in c or "wgrad" in c or "presplit" in c))
        flags.WGRAD_F16X3 = flags.BN_BWD_PLANES = flags.BN_ACT_PLANES = False
        y_old, g_old = run()
    finally:
        _abi.check = real
        flags.WGRAD_F16X3 = flags.BN_BWD_PLANES = flags.BN_ACT_PLANES = True
    # float64 reference
    x64 = [t.detach().double().requires_grad_(want_x) for t in blocks]
    h = torch.cat(x64, 1)
    p64 = []
    for i in range(n_stage):
        w = lins[i].weight.detach().double().requires_grad_(True); p64.append(w)
        h = h @ w.t()
        if lins[i].bias is not None:
            b = lins[i].bias.detach().double().requires_grad_(True); p64.append(b)
            h = h + b
        bn = bns[i]
        if bn is not None:
            if modes[i] == "train":
                mean, var = h.mean(0), h.var(0, unbiased=False)
            else:
                mean, var = bn.running_mean.double(), bn.running_var.double()
            g_, b_ = bn.weight.detach().double(), bn.bias.detach().double()
            if bn.weight.requires_grad:
                g_.requires_grad_(True); b_.requires_grad_(True)
            h = (h - mean) / torch.sqrt(var + bn.eps) * g_ + b_
            bns[i]._g64 = (g_, b_)
        h = ACTS[acts[i]](h)
    (h * wout.double()).sum().backward()
    ref = ([t.grad for t in x64] if want_x else [])
    for i in range(n_stage):
        ref.append([p for p in p64 if True][0]); p64.pop(0)
        ref[-1] = ref[-1].grad
        if lins[i].bias is not None:
            ref.append(p64.pop(0).grad)
    for i in range(n_stage):
        if bns[i] is not None and bns[i].weight.requires_grad:
            ref += [bns[i]._g64[0].grad, bns[i]._g64[1].grad]
    # parameter order of `params`: all Linear parameters first (weight, bias per stage), then the BatchNorm ones -- as built above
    worst = 0.0
    ey_new = float((y_new.double() - h.detach()).abs().max() / h.detach().abs().max())
    ey_old = float((y_old.double() - h.detach()).abs().max() / h.detach().abs().max())
    assert ey_new <= max(2.0 * ey_old, 2e-6), ("forward", ey_new, ey_old)
    for k, (gn, go, r) in enumerate(zip(g_new, g_old, ref)):
        scale = float(r.abs().max())
        if scale == 0.0:
            continue
        e_new, e_old = float((gn.double() - r).abs().max()) / scale, float((go.double() - r).abs().max()) / scale
        if e_old < 1e-3:                            # (a bias in front of a train-mode BatchNorm: its true gradient is zero, both runs return rounding noise)
            assert e_new <= max(3.0 * e_old, 3e-6), ("gradient %d" % k, e_new, e_old, tuple(r.shape))
        worst = max(worst, e_new if e_old < 1e-3 else 0.0)
    return worst, used


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    worst, seen = 0.0, {}
    for seed in range(first, first + n):
        try:
            w, used = one_case(seed)
        except Exception:
            print("FAILED seed %d: %s" % (seed, getattr(one_case, "desc", "?")), flush=True)
            raise
        worst = max(worst, w)
        for u in used:
            seen[u] = seen.get(u, 0) + 1
    print("soak_planes: %d cases from seed %d ok; worst gradient error over the largest magnitude %.2e; entries used: %s" % (n, first, worst, seen))


if __name__ == "__main__":
    main()

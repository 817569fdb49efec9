#!/usr/bin/env python3
"""Soak run of the GRADIENTS of the layer classes against PyTorch autograd over the oracle's restatement, fresh seeds (test
infrastructure, not collected by pytest): random class, message kind, identifier scope, flow, widths, central-encoder kinds
(one-hot / embedding, extended or not), eps trained or fixed, BatchNorm on / off, train OR eval mode (eval: the fused forward + the
recompute on the HIP adjoints, tests/test_eval_grad_gpu.py), 1-40 graphs incl. edge-less ones.  Smooth activations (elu / tanh /
identity) so that two correct implementations cannot land on different sides of a kink; the ogb classes (relu by definition) are run
at small sizes with the comparison restricted to cases whose pre-activations keep a margin from zero in the oracle.

    python tests/soak_grads.py [first_seed] [n_seeds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsn_amd import layers, synth  # noqa: E402
from oracle import oracle  # noqa: E402


def one_case(seed):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    cls = str(rng.choice(["GSN_sparse", "GSN_edge_sparse", "MPNN_sparse", "MPNN_edge_sparse", "GSN_edge_sparse_ogb", "MPNN_edge_sparse_ogb"]))
    ogb = cls.endswith("_ogb")
    has_ids, has_ef = cls.startswith("GSN"), "edge" in cls
    kind = "ogb" if ogb else str(rng.choice(["general", "gin"]))
    scope = str(rng.choice(["local", "global"]))
    flow = str(rng.choice(["source_to_target", "target_to_source"]))
    bn = bool(rng.random() < 0.7)
    act = "relu" if ogb else str(rng.choice(["elu", "tanh", "identity"]))
    training = bool(rng.random() < 0.5)
    d = int(rng.choice([int(rng.integers(2, 20)), int(rng.integers(2, 12)) * 4, 32]))
    ctor = dict(d_degree=1, degree_as_tag=False, retain_features=True, seed=seed, activation_name=act, bn=bn, flow=flow, aggr="add",
                eps=float(rng.choice([0.0, 0.25, -0.5])), train_eps=bool(rng.random() < 0.6),
                extend_dims=bool(rng.random() < 0.7), id_embedding=str(rng.choice(["one_hot_encoder", "embedding"])),
                edge_embedding=str(rng.choice(["one_hot_encoder", "embedding"])))
    if ogb:
        d_x = d_id = d_ef = d
        ctor.update(d_in=d, d_msg=None, d_up=int(rng.integers(2, 24)), d_h=[int(rng.integers(2, 40))], msg_kind="ogb")
        if has_ids:
            ctor.update(d_id=d, id_scope=scope)
        ctor["d_ef"] = d
    else:
        d_x, d_id, d_ef = int(rng.integers(1, 20)), int(rng.integers(1, 9)), int(rng.integers(1, 7))
        ctor.update(d_in=d_x, d_msg=(None if kind == "gin" else int(rng.integers(2, 24))), d_up=d,
                    d_h=[int(rng.integers(2, 40))] if rng.random() < 0.8 else [], msg_kind=kind)
        if has_ids:
            ctor.update(d_id=d_id, id_scope=scope)
        if has_ef:
            ctor["d_ef"] = d_ef
    n_graphs = int(rng.integers(1, 41))
    b = synth.zinc_shape_batch(n_graphs, seed=seed, mean_n=float(rng.choice([3.0, 9.0, 23.0])), sd_n=3.0, n_min=1, n_max=40)
    n, E = b.num_nodes, b.num_edges
    ei = torch.from_numpy(b.edge_index)
    if rng.random() < 0.08:                        # an edge-less batch
        ei, E = ei[:, :0].contiguous(), 0
    one_case.desc = "%s %s %s %s bn=%d act=%s train=%d d=%d n=%d E=%d d_h=%s" % (cls, kind, scope, flow, bn, act, training, d, n, E, ctor["d_h"])
    layer = getattr(layers, cls)(**ctor)
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.3, 0.3)
    layer.train(training)
    if training and bn and n < 2:
        return None
    x = torch.randn(n, d_x)
    kw = {}
    if has_ids:
        kw["identifiers"] = torch.randn(E if scope == "local" else n, d_id)
    if has_ef:
        kw["edge_features"] = torch.randn(E, d_ef)
    if training and bn and E == 1 and kind == "general":      # (one edge row through a train-mode BatchNorm: the reference raises; E == 0 is legal)
        return None
    pn = {k for k, _ in layer.named_parameters()}
    sd = {k: v.clone().requires_grad_(k in pn) for k, v in layer.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    kwr = {k: v.clone().requires_grad_(True) for k, v in kw.items()}
    yr = oracle.layer_forward(cls, ctor, sd, xr, ei, degrees=None, training=training, **kwr)
    w = torch.randn_like(yr)
    (yr * w).sum().backward()
    layer.cuda()
    xg = x.cuda().requires_grad_(True)
    kwg = {k: v.cuda().requires_grad_(True) for k, v in kw.items()}
    extra = {} if (has_ids or not ogb) else {"identifiers": None}
    y = layer(xg, ei.cuda(), degrees=torch.zeros(n, device="cuda"), **kwg, **extra)
    (y * w.cuda()).sum().backward()
    tol = 2e-3 if ogb else 5e-5                     # (relu: a pre-activation within rounding of zero flips a whole gradient row)
    desc = "%s %s %s %s bn=%d act=%s train=%d d=%d n=%d E=%d" % (cls, kind, scope, flow, bn, act, training, d, n, E)
    ymax = max(float(yr.detach().abs().max()), 1e-20)
    errs = {"y": float((y.detach().cpu() - yr.detach()).abs().max()) / ymax}
    gmax = max([float(xr.grad.abs().max())] + [float(v.grad.abs().max()) for v in sd.values() if v.requires_grad and v.grad is not None] + [1e-20])
    errs["dx"] = float((xg.grad.cpu() - xr.grad).abs().max()) / gmax
    for k, v in kwr.items():
        if v.grad is not None and v.numel():
            errs["d" + k] = float((kwg[k].grad.cpu() - v.grad).abs().max()) / gmax
    got = dict(layer.named_parameters())
    for k, v in sd.items():
        if v.requires_grad and v.grad is not None:
            if got[k].grad is None:
                errs["missing " + k] = 1.0
            else:
                errs[k] = float((got[k].grad.cpu() - v.grad).abs().max()) / gmax
    bad = {k: e for k, e in errs.items() if not (e <= (1e-5 if k == "y" else tol))}
    return desc, bad


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    fails = cases = 0
    for seed in range(first, first + n_seeds):
        try:
            r = one_case(seed)
        except Exception as ex:        # (a crash is a failure with its seed)
            fails += 1
            print("seed %d raised %s: %s :: %s" % (seed, type(ex).__name__, str(ex)[:200], getattr(one_case, "desc", "")))
            continue
        if r is None:
            continue
        cases += 1
        desc, bad = r
        if bad:
            fails += 1
            print("seed %d FAIL %s :: %s" % (seed, desc, {k: "%.2e" % v for k, v in bad.items()}))
    print("gradient soak: %d cases, %d failures" % (cases, fails))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())

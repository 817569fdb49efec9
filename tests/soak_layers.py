#!/usr/bin/env python3
"""Soak run of the layer classes against the oracle's fp32 restatement with fresh seeds (test infrastructure, not collected by
pytest): random class, message kind, identifier scope, flow, widths (any integer, not only the reference's 64 / 128 / 300), input
widths, BatchNorm on / off, activation, eval / train-mode forward, 1-700 graphs per batch (so that single-tile and multi-tile
launches both occur).  Tolerance 1e-5 of the largest output (element-wise with the row-max floor of test_layers_gpu.py).

    python tests/soak_layers.py [first_seed] [n_seeds] [--wide]     (--wide: the `general` cases take the d = 128 shape of csrc/layer_w.hip)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsn_amd import layers, synth  # noqa: E402
from oracle import oracle  # noqa: E402


WIDE = "--wide" in sys.argv
if WIDE:
    sys.argv.remove("--wide")


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    fails = cases = 0
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        torch.manual_seed(seed)
        cls = str(rng.choice(["GSN_sparse", "GSN_edge_sparse", "MPNN_sparse", "MPNN_edge_sparse", "GSN_edge_sparse_ogb", "MPNN_edge_sparse_ogb"]))
        ogb = cls.endswith("_ogb")
        has_ids, has_ef = cls.startswith("GSN"), "edge" in cls
        kind = "ogb" if ogb else str(rng.choice(["general", "gin"]))
        if WIDE and not ogb:
            kind = "general"
        d = int(rng.choice([int(rng.integers(3, 40)), int(rng.integers(8, 50)) * 4, 64, 128])) if not ogb else int(rng.choice([int(rng.integers(4, 80)) * 4, 64, 300]))
        scope = str(rng.choice(["local", "global"]))
        flow = str(rng.choice(["source_to_target", "target_to_source"]))
        bn = bool(rng.random() < 0.7)
        act = str(rng.choice(["relu", "elu", "tanh", "identity"])) if not ogb else "relu"
        training = bool(rng.random() < 0.35) and bn
        ctor = dict(d_degree=1, degree_as_tag=False, retain_features=True, seed=seed, activation_name=act, bn=bn, flow=flow, aggr="add", eps=0,
                    extend_dims=True, id_embedding="one_hot_encoder", edge_embedding="one_hot_encoder", train_eps=bool(rng.random() < 0.5))
        if ogb:
            d_x = d_id = d_ef = d
            ctor.update(d_in=d, d_id=d, id_scope=scope if has_ids else "local", d_msg=None, d_up=d, d_h=[2 * d], msg_kind="ogb")
            if has_ef:
                ctor["d_ef"] = d
        elif kind == "general" and WIDE:
            # the d = 128 hidden-layer shape of csrc/layer_w.hip (all widths 128, <= 16 per-edge / per-end-point columns in multiples of 4), eval
            d = d_x = 128
            d_id, d_ef = int(rng.choice([4, 8, 12] if scope == "local" or not has_ef else [4])), 4
            if scope == "global" and has_ef:
                d_id, d_ef = 4, 8
            if scope == "global" and not has_ef:
                d_id = int(rng.choice([4, 8]))
            act = str(rng.choice(["relu", "identity"]))
            training = False
            ctor.update(activation_name=act, d_in=128, d_id=d_id, id_scope=scope, d_msg=128, d_up=128, d_h=[128], msg_kind="general")
            if has_ef:
                ctor["d_ef"] = d_ef
        else:
            d_x, d_id, d_ef = int(rng.integers(1, 40)), int(rng.integers(1, 20)), int(rng.integers(1, 9))
            if kind == "gin":
                d_x = d
            ctor.update(d_in=d_x, d_id=d_id, id_scope=scope, d_msg=(None if kind == "gin" else int(rng.integers(4, 130))), d_up=d,
                        d_h=[int(rng.integers(4, 140))], msg_kind=kind)
            if has_ef:
                ctor["d_ef"] = d_ef
        G = int(rng.choice([1, 3, 40, 200, 700]))
        b = synth.zinc_shape_batch(G, seed=seed)
        N, E = b.num_nodes, b.num_edges
        try:
            layer = getattr(layers, cls)(**ctor)
        except Exception as ex:                                   # a combination the reference's constructor refuses as well
            print("skip seed %d (%s): %s" % (seed, cls, str(ex)[:80]))
            continue
        for m in layer.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.3, 0.3)
        layer.train(training)
        x = torch.randn(N, d_x)
        if WIDE:                                                  # rows of very different magnitude: every edge row has its own scale
            x = x * torch.exp2(torch.randint(-12, 13, (N, 1)).float())
        ei = torch.from_numpy(b.edge_index)
        ids_scope = ctor["id_scope"]
        ids = torch.randn(E if ids_scope == "local" else N, d_id) * 0.5 if has_ids else None
        ef = torch.randn(E, d_ef) * 0.5 if has_ef else None
        sd = {k: v.clone() for k, v in layer.state_dict().items()}
        what = (cls, kind, ids_scope, flow, "d", d, "bn", bn, act, "train", training, "G", G, {k: ctor[k] for k in ("d_in", "d_msg", "d_h") if k in ctor})
        try:
            ref = oracle.layer_forward(cls, ctor, sd, x, ei, identifiers=ids, degrees=None, edge_features=ef, training=training)
        except Exception as ex:
            print("skip seed %d (oracle): %s %s" % (seed, what, str(ex)[:80]))
            continue
        layer.cuda()
        kw = dict(degrees=torch.zeros(N, device="cuda"), identifiers=None if ids is None else ids.cuda())
        if has_ef:
            kw["edge_features"] = ef.cuda()
        with torch.no_grad():
            y = layer(x.cuda(), ei.cuda(), **kw).cpu()
        cases += 1
        ok = y.shape == ref.shape
        if ok:
            err = (y - ref).abs()
            tol = 1e-5 * ref.abs() + 1e-5 * ref.abs().amax(dim=1, keepdim=True) + 1e-6 * float(ref.abs().max())
            ok = bool((err <= tol).all()) and not bool(torch.isnan(y).any())
        if not ok:
            fails += 1
            print("FAIL seed %d %s: max err %.2e of max %.2e" % (seed, what, float((y - ref).abs().max()) if y.shape == ref.shape else -1, float(ref.abs().max())), flush=True)
    print("layer soak: %d cases, %d failures" % (cases, fails))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())

"""The reference-generated 5 x 300 + virtual-node train step (tests/golden/model_ogb300.npz) under the dense-stage modes:
worst gradient-digest deviation from the golden (bar 1e-4) and the five worst keys.  Usage: python scripts/gpu/ogb300_modes.py
(spawns itself once per mode: the modes are read from the environment at import)."""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one():
    import types
    import numpy as np
    import torch
    import helpers
    from gsn_amd import models
    DEV = "cuda:0"
    z = np.load(os.path.join(ROOT, "tests", "golden", "model_ogb300.npz"), allow_pickle=False)
    L, dm = 5, 300
    atom_dims, bond_dims, id_dims = [119, 4, 12, 12, 10, 6, 6, 2, 2], [5, 6, 2], [3, 3, 3, 3]
    kw = dict(seed=0, model_name="GSN_edge_sparse_ogb", readout="mean", dropout_features=[0.0] * (L + 1), bn=[True] * L,
              final_projection=[False] * L + [True], residual=False, inject_ids=True, vn=True, id_scope="local",
              d_msg=[dm] * L, d_out=[dm] * L, d_h=[[2 * dm]] * L, aggr="add", flow="source_to_target", msg_kind="ogb",
              train_eps=[True] * L, activation_mlp="relu", bn_mlp=True, jk_mlp=False, degree_embedding="None",
              degree_as_tag=[False] * L, retain_features=[True] * L, multi_embedding_aggr="sum", features_scope="full",
              input_node_encoder="embedding", d_out_node_encoder=dm, input_vn_encoder="embedding", d_out_vn_encoder=dm,
              edge_encoder="embedding", d_out_edge_encoder=[dm] * L, id_embedding="embedding", d_out_id_embedding=dm,
              d_out_degree_embedding=dm, d_out_vn=[dm] * (L - 1), vn_pooling="sum", extend_dims=True, activation="relu")
    model = models.GNN_OGB(len(atom_dims), 1, None, id_dims, len(bond_dims), atom_dims, bond_dims, None, None, **kw)
    keys = [str(k) for k in z["shape_keys"]]
    ptr, flat = z["shape_ptr"], z["shape_flat"]
    shapes = {k: tuple(int(d) for d in flat[ptr[i]:ptr[i + 1]]) for i, k in enumerate(keys)}
    model.load_state_dict(helpers.procedural_state(shapes))
    model = model.to(DEV).train(True)
    data = types.SimpleNamespace(**{a: torch.from_numpy(z["data/" + a]).to(DEV) for a in ("x", "edge_index", "identifiers", "batch", "degrees", "edge_features")})
    pred = model(data)
    want = torch.from_numpy(z["pred"])
    prel = float((pred.detach().cpu() - want).abs().max() / want.abs().max())
    (pred * torch.from_numpy(z["gy"]).to(DEV)).sum().backward()
    dg = helpers.grad_digest({k: p.grad for k, p in model.named_parameters() if p.grad is not None})
    gkeys = [str(k) for k in z["grad_keys"]]
    gmax = float(z["grad_digest"][:, 1].max())
    dev = []
    for k, (proj, nrm, mx) in zip(gkeys, z["grad_digest"]):
        p_, n_, m_ = dg[k]
        scale = max(nrm, (3e-2 if int(np.prod(shapes[k])) <= 1 else 1e-4) * gmax)
        dev.append((max(abs(p_ - proj), abs(n_ - nrm)) / scale, k))
    dev.sort(reverse=True)
    print("mode", os.environ.get("GSN_MODE_NAME"), "pred rel %.3g" % prel, "worst digest %.3g" % dev[0][0],
          "median %.3g" % dev[len(dev) // 2][0], "top:", ["%s %.2g" % (k.replace("GNN_layers", "L"), d) for d, k in dev[:5]], flush=True)


if __name__ == "__main__":
    if os.environ.get("GSN_MODE_NAME"):
        one()
    else:
        modes = {"bf16x6": {}, "f16x3_stats": {"GSN_LINEAR_F16X3_STATS": "1"}}
        for extra in sys.argv[1:]:          # NAME:VAR=VAL,VAR=VAL
            name, kv = extra.split(":")
            modes[name] = dict(p.split("=") for p in kv.split(","))
        for name, env in modes.items():
            e = dict(os.environ, GSN_MODE_NAME=name, **env)
            subprocess.call([sys.executable, os.path.abspath(__file__)], env=e)

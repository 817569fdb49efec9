#!/usr/bin/env python3
"""Layer-only timing of the `general` layer forward (BASELINE config 2, layer 0) at the bench shape: the one-launch kernel
(gsn_layer_fused_fwd_hip) next to the multi-launch path, HIP events around the launches.  SURVEY 8(d): B_alg, F_alg."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gsn_amd import flags, layers, synth  # noqa: E402

CTOR = dict(d_in=28, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=128,
            d_up=128, d_h=[128], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--float-inputs", action="store_true", help="real-valued inputs (no row is exact in fp16)")
    ap.add_argument("--wide", action="store_true", help="a hidden layer of the d = 128 model: d_in = 128, K = 272 edge rows (csrc/layer_w.hip)")
    args = ap.parse_args()
    b = synth.zinc_shape_batch(args.graphs, seed=1000)
    N, E = b.num_nodes, b.num_edges
    dev = "cuda"
    x = torch.nn.functional.one_hot(torch.from_numpy(b.atom_type), 28).float().to(dev)
    ef = torch.nn.functional.one_hot(torch.from_numpy(b.bond_type), 4).float().to(dev)
    ids = torch.nn.functional.one_hot(torch.randint(0, 3, (E, 4)), 3).reshape(E, 12).float().to(dev)
    if args.float_inputs:
        x, ef, ids = torch.randn_like(x), torch.randn_like(ef), torch.randn_like(ids)
    d_in = 28
    if args.wide:
        d_in = 128
        x = torch.randn(N, 128, device=dev).relu()
    ei = torch.from_numpy(b.edge_index).to(dev)
    deg = torch.zeros(N, device=dev)
    torch.manual_seed(0)
    layer = layers.GSN_edge_sparse(**dict(CTOR, d_in=d_in)).to(dev).eval()
    b_alg = 16.0 * E + 4.0 * (N * d_in + E * 12 + E * 4 + N * 128)
    f_alg = 2.0 * E * ((2 * d_in + 16) * 128 + 128 * 128) + 2.0 * N * ((d_in + 128) * 128 + 128 * 128)
    res = {"graphs": args.graphs, "N": N, "E": E, "B_alg_bytes": b_alg, "F_alg_flops": f_alg}
    ys = {}
    from gsn_amd import packs
    variants = [("fused", True), ("multi_launch", False)]
    if not args.float_inputs and not args.wide:
        variants.insert(0, ("fused_pack16", True))       # tagged exact inputs: csrc/layer_rp.hip
    for name, fused in variants:
        flags.FUSED_LAYER = fused
        if name == "fused_pack16":
            packs.node_pack(x)
            packs.edge_pack([ids, ef])
        else:
            for t in (x, ids, ef):
                packs.release(t)
        with torch.no_grad():
            for _ in range(10):
                y = layer(x, ei, identifiers=ids, degrees=deg, edge_features=ef)      # CSR cached: the layer launches only
            torch.cuda.synchronize()
            flags.KERNEL_TIMER = {}
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                y = layer(x, ei, identifiers=ids, degrees=deg, edge_features=ef)
            e1.record()
            torch.cuda.synchronize()
            timer, flags.KERNEL_TIMER = flags.KERNEL_TIMER, None
        ms = e0.elapsed_time(e1) / args.steps
        per = {k: round(sum(a.elapsed_time(bb) for a, bb, _ in v) / args.steps, 4) for k, v in timer.items()}
        res[name] = {"ms_per_layer": round(ms, 4), "hbm_frac_of_B_alg": round(b_alg / (ms * 1e-3) / 8e12, 4), "kernels_ms": per}
        ys[name] = y
    err = float((ys["fused"] - ys["multi_launch"]).abs().max() / ys["multi_launch"].abs().max())
    res["max_diff_over_max"] = float("%.3g" % err)
    if "fused_pack16" in ys:
        res["pack16_max_diff_over_max"] = float("%.3g" % float((ys["fused_pack16"] - ys["multi_launch"]).abs().max() / ys["multi_launch"].abs().max()))
    print(json.dumps(res))


if __name__ == "__main__":
    main()

"""Graph-aligned d = 128 layer kernel (csrc/layer_g.hip) against the oracle and against csrc/layer_w.hip; timing of both at the bench size.
usage: python scripts/gpu/g_check.py [--time] [--graphs 65536]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))

from gsn_amd import flags, layers, synth  # noqa: E402
from oracle import oracle  # noqa: E402
from test_fused_gpu import WIDE, _wide_ctor, _randomise_bn, _elementwise_ok  # noqa: E402


def run_case(cls, ctor, b, x, ids, ef, seed, ref=True, partition=True):
    ei = torch.from_numpy(b.edge_index)
    torch.manual_seed(seed)
    layer = getattr(layers, cls)(**ctor)
    _randomise_bn(layer, seed + 1)
    layer.eval()
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    kw = dict(identifiers=ids, degrees=None)
    if ef is not None:
        kw["edge_features"] = ef
    yref = oracle.layer_forward(cls, ctor, sd, x, ei, training=False, **kw) if ref else None
    layer.cuda()
    eic = ei.cuda()
    kwg = dict(identifiers=None if ids is None else ids.cuda(), degrees=torch.zeros(x.shape[0], device="cuda"))
    if ef is not None:
        kwg["edge_features"] = ef.cuda()
    if partition:
        mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
        assert layers.set_graph_partition(eic, torch.from_numpy(b.node_ptr).cuda(), torch.from_numpy(b.edge_ptr).cuda(), mn, me)
    outs = {}
    for name, flag in (("g", True), ("w", False)):
        flags.GRAPH_ALIGNED_LAYER = flag
        layers._CSR_CACHE.clear()
        with torch.no_grad():
            outs[name] = layer(x.cuda(), eic, **kwg).cpu()
    flags.GRAPH_ALIGNED_LAYER = True
    return outs, yref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--graphs", type=int, default=65536)
    ap.add_argument("--skip-checks", action="store_true")
    args = ap.parse_args()
    os.environ["GSN_CHAIN_TRACE"] = "1"
    ok_all = True
    if not args.skip_checks:
        for n_graphs in (1, 7, 300):
            for cls, ctor_kw, d_id, d_ef in WIDE:
                b = synth.zinc_shape_batch(n_graphs, seed=41 + n_graphs)
                g = torch.Generator().manual_seed(13)
                N, E = b.num_nodes, b.num_edges
                x = torch.randn(N, 128, generator=g).relu()
                ctor = _wide_ctor(cls, ctor_kw)
                ids = torch.randn(N if ctor.get("id_scope") == "global" else E, d_id, generator=g).abs() if d_id else None
                ef = torch.randn(E, d_ef, generator=g) if d_ef else None
                outs, ref = run_case(cls, ctor, b, x, ids, ef, seed=9)
                okg, okw = _elementwise_ok(outs["g"], ref), _elementwise_ok(outs["w"], ref)
                eg = float((outs["g"] - ref).abs().max() / ref.abs().max())
                ew = float((outs["w"] - ref).abs().max() / ref.abs().max())
                print("case %-18s id_scope %-6s graphs %4d: g ok %s (%.2e)  w ok %s (%.2e)" % (cls, ctor.get("id_scope"), n_graphs, okg, eg, okw, ew), flush=True)
                ok_all &= okg
        # mixed magnitudes, hubs inside <= 128-node graphs, isolated nodes, edge-less graphs, both flows
        rng = np.random.default_rng(5)
        graphs = [(5, np.zeros((2, 0), dtype=np.int64)), synth.er_graph(40, 300, 1)]
        star = np.stack([np.zeros(100, dtype=np.int64), np.arange(1, 101)])
        graphs.append((128, np.concatenate([star, star[::-1]], axis=1)))
        graphs += [synth.zinc_shape_graph(rng) for _ in range(40)]
        graphs.append(synth.er_graph(128, 1000, 2))
        graphs.append((70, np.zeros((2, 0), dtype=np.int64)))
        graphs += [(1, np.zeros((2, 0), dtype=np.int64)) for _ in range(150)]
        graphs += [synth.zinc_shape_graph(rng) for _ in range(11)]
        b = synth.collate(graphs)
        g = torch.Generator().manual_seed(19)
        N, E = b.num_nodes, b.num_edges
        x = torch.randn(N, 128, generator=g) * torch.exp2(torch.randint(-20, 21, (N, 1), generator=g).float())
        x[::7] *= torch.logspace(-4, 0, 128)
        ef = torch.randn(E, 4, generator=g) * torch.exp2(torch.randint(-10, 11, (E, 1), generator=g).float())
        ids = torch.randint(0, 3, (E, 12), generator=g).float()
        cls, ctor_kw = WIDE[0][0], WIDE[0][1]
        for flow in ("source_to_target", "target_to_source"):
            outs, ref = run_case(cls, _wide_ctor(cls, ctor_kw, flow=flow), b, x, ids, ef, seed=10)
            okg = _elementwise_ok(outs["g"], ref)
            print("mixed magnitudes / hubs, %s: g ok %s (%.2e) w ok %s" % (flow, okg, float((outs["g"] - ref).abs().max() / ref.abs().max()),
                                                                        _elementwise_ok(outs["w"], ref)), flush=True)
            if not okg:
                d = (outs["g"] - ref).abs() - 1e-5 * ref.abs() - 1e-5 * ref.abs().amax(dim=1, keepdim=True)
                rows = torch.nonzero((d > 0).any(dim=1)).flatten()
                print("  failing rows", rows[:20].tolist(), "of", N, "node_ptr", b.node_ptr[:8])
            ok_all &= okg
        # non-finite
        b = synth.zinc_shape_batch(40, seed=77)
        g = torch.Generator().manual_seed(23)
        N, E = b.num_nodes, b.num_edges
        x = torch.randn(N, 128, generator=g)
        ef = torch.randn(E, 4, generator=g)
        ids = torch.randn(E, 12, generator=g)
        x[17, 5] = float("inf")
        ef[33, 2] = float("nan")
        outs, ref = run_case(cls, _wide_ctor(cls, ctor_kw), b, x, ids, ef, seed=11)
        y = outs["g"]
        bad_ref = ~torch.isfinite(ref).all(dim=1)
        bad = ~torch.isfinite(y).all(dim=1)
        okn = torch.equal(bad, bad_ref) and bool(bad.any()) and bool(torch.isnan(y[bad]).all()) and _elementwise_ok(y[~bad], ref[~bad])
        print("non-finite: ok %s  (bad rows %d, ref %d)" % (okn, int(bad.sum()), int(bad_ref.sum())), flush=True)
        ok_all &= okn
        print("ALL OK" if ok_all else "FAILURES", flush=True)
    if args.time:
        os.environ.pop("GSN_CHAIN_TRACE", None)
        dev = torch.device("cuda")
        b = synth.zinc_shape_batch(args.graphs, seed=1000)
        N, E = b.num_nodes, b.num_edges
        g = torch.Generator().manual_seed(6)
        x = torch.randn(N, 128, generator=g).relu().to(dev)
        ids = torch.randn(E, 12, generator=g).abs().to(dev)
        ef = torch.randn(E, 4, generator=g).to(dev)
        deg = torch.zeros(N, device=dev)
        ei = torch.from_numpy(b.edge_index).to(dev)
        mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
        layers.set_graph_partition(ei, torch.from_numpy(b.node_ptr).to(dev), torch.from_numpy(b.edge_ptr).to(dev), mn, me)
        torch.manual_seed(0)
        ctor = dict(d_in=128, d_ef=4, d_id=12, d_degree=1, degree_as_tag=False, retain_features=True, id_scope="local", d_msg=128, d_up=128,
                    d_h=[128], seed=0, activation_name="relu", bn=True, msg_kind="general", flow="source_to_target")
        layer = layers.GSN_edge_sparse(**ctor).to(dev).eval()
        res = {}
        for name, flag in (("g", True), ("w", False), ("g", True), ("w", False)):
            flags.GRAPH_ALIGNED_LAYER = flag
            with torch.no_grad():
                for _ in range(20):
                    y = layer(x, ei, identifiers=ids, degrees=deg, edge_features=ef)
                torch.cuda.synchronize()
                flags.KERNEL_TIMER = {}
                for _ in range(20):
                    y = layer(x, ei, identifiers=ids, degrees=deg, edge_features=ef)
                torch.cuda.synchronize()
            evs = flags.KERNEL_TIMER.get("layer_fused", [])
            flags.KERNEL_TIMER = None
            ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs) / max(len(evs), 1)
            res.setdefault(name, []).append(round(ms, 4))
            res[name + "_y"] = y
        print("graphs %d N %d E %d: layer_g %s ms, layer_w %s ms" % (args.graphs, N, E, res["g"], res["w"]), flush=True)
        print("g vs w element-wise:", _elementwise_ok(res["g_y"].cpu(), res["w_y"].cpu()), flush=True)
        flags.GRAPH_ALIGNED_LAYER = True


if __name__ == "__main__":
    main()

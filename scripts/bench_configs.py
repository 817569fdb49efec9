#!/usr/bin/env python3
"""Secondary measurements for the other BASELINE.json configs (not the headline bench): counting throughput on
SR25-like / clique-rich / ER-128 inputs and forward(+backward) times of the gin / ogb / train-mode layer paths.
Synthetic inputs only (the GPU box has no reference data).  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch
import networkx as nx

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsn_amd import flags, synth, layers  # noqa: E402
from gsn_amd.counting import CountPlan, count_batch  # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def count_case(name, graphs, pats, mode, induced, reps=3):
    b = synth.collate(graphs)
    plan = CountPlan.get(pats, mode, induced)
    dev = torch.device("cuda")
    node_ptr = torch.from_numpy(b.node_ptr).to(dev); edge_ptr = torch.from_numpy(b.edge_ptr).to(dev)
    ei = torch.from_numpy(b.edge_index).to(dev)
    mn, me = int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max())
    rows = b.num_edges if mode == "edge" else b.num_nodes
    out = torch.empty((rows, plan.n_cols), dtype=torch.int64, device=dev)
    dt = timeit(lambda: count_batch(plan, node_ptr, edge_ptr, ei, max_nodes=mn, max_edges=me, device=dev, out=out, check=False), reps=reps, warm=1)
    occ = int(out.sum())
    return {"case": name, "graphs": b.num_graphs, "mode": mode, "induced": induced, "cols": plan.n_cols, "ms": round(dt * 1e3, 3),
            "graphs_per_s": round(b.num_graphs / dt, 1), "sum_counts": occ}


def clique_union_graph(rng, n, n_cliques, size):
    """IMDB-like ego network: union of random cliques + a hub."""
    adj = np.zeros((n, n), bool)
    for _ in range(n_cliques):
        m = rng.choice(n, size=min(size, n), replace=False)
        adj[np.ix_(m, m)] = True
    adj[0, :] = adj[:, 0] = True
    np.fill_diagonal(adj, False)
    und = np.argwhere(np.triu(adj, 1))
    return n, synth.undirected_to_edge_index(n, und)


def main():
    res = {"counting": [], "layers": []}
    rng = np.random.default_rng(0)
    cyc = lambda ks: [list(nx.cycle_graph(k).edges) for k in ks]
    clq = lambda ks: [list(nx.complete_graph(k).edges) for k in ks]
    # config 1 shape: 25-vertex 12-regular graphs x 15 (random regular stands in for SR(25,12,5,6))
    sr = []
    for s in range(15):
        g = nx.random_regular_graph(12, 25, seed=s)
        sr.append((25, synth.undirected_to_edge_index(25, list(g.edges()))))
    res["counting"].append(count_case("12-regular n=25 x15, cycle 3..6 induced (config 1 shape)", sr, cyc(range(3, 7)), "edge", True))
    # config 2 dataset size
    zb = synth.zinc_shape_batch(12000, seed=0)
    res["counting"].append(count_case("ZINC-shape x12000, cycle 3..6 (config 2 dataset)", [zb.graph(i) for i in range(12000)], cyc(range(3, 7)), "edge", False))
    # config 3 shape
    imdb = [clique_union_graph(rng, int(rng.integers(12, 60)), int(rng.integers(1, 4)), int(rng.integers(6, 14))) for _ in range(1000)]
    res["counting"].append(count_case("clique-union ego nets x1000, complete 3..5 (config 3 shape)", imdb, clq(range(3, 6)), "vertex", False))
    # config 5 shape
    g5 = []
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "orbits.npz")
    z = np.load(path)
    pats5 = [z["all_simple_graphs_5/%d/edges" % i].tolist() for i in range(21)]
    er = [synth.er_graph(128, 1000, s) for s in range(256)]
    for mode in ("vertex", "edge"):
        res["counting"].append(count_case("ER G(128,1000) x256, 21 five-vertex patterns (config 5)", er, pats5, mode, False, reps=2))

    # layers: gin (config 3), ogb (config 4), general train mode fwd+bwd (config 2 training)
    dev = "cuda"
    b = synth.zinc_shape_batch(4096, seed=1)
    b = __import__("bench").make_batch(65536, 5) if False else b
    N, E = b.num_nodes, b.num_edges
    ei = torch.from_numpy(b.edge_index).to(dev)

    def layer_case(name, layer, x, kw, train, backward):
        layer = layer.to(dev).train(train)
        xs = x.clone().requires_grad_(backward)

        def run():
            y = layer(xs, ei, **kw)
            if backward:
                y.sum().backward()
        with torch.set_grad_enabled(backward):
            dt = timeit(run, reps=5, warm=2)
        res["layers"].append({"case": name, "graphs": b.num_graphs, "N": N, "E": E, "train": train, "backward": backward,
                              "ms": round(dt * 1e3, 3), "graphs_per_s": round(b.num_graphs / dt, 1)})

    torch.manual_seed(0)
    base = dict(d_degree=1, degree_as_tag=False, retain_features=True, seed=0, activation_name="relu", bn=True)
    x64 = torch.randn(N, 64, device=dev)
    layer_case("GSN_sparse gin/global d=64 fwd (config 3 layer)", layers.GSN_sparse(d_in=64, d_id=40, id_scope="global", d_msg=None, d_up=64,
               d_h=[64], msg_kind="gin", train_eps=True, id_embedding="one_hot_encoder", extend_dims=True, **base), x64,
               {"identifiers": torch.randn(N, 40, device=dev), "degrees": torch.zeros(N, device=dev)}, False, False)
    x300 = torch.randn(N, 300, device=dev)
    layer_case("GSN_edge_sparse_ogb d=300 fwd (config 4 layer)", layers.GSN_edge_sparse_ogb(d_in=300, d_ef=300, d_id=300, id_scope="local",
               d_msg=None, d_up=300, d_h=[600], msg_kind="ogb", train_eps=True, **base), x300,
               {"identifiers": torch.randn(E, 300, device=dev), "degrees": torch.zeros(N, device=dev), "edge_features": torch.randn(E, 300, device=dev)}, False, False)
    layer_case("GSN_edge_sparse_ogb d=300 train fwd+bwd (config 4 step)", layers.GSN_edge_sparse_ogb(d_in=300, d_ef=300, d_id=300, id_scope="local",
               d_msg=None, d_up=300, d_h=[600], msg_kind="ogb", train_eps=True, **base), x300,
               {"identifiers": torch.randn(E, 300, device=dev), "degrees": torch.zeros(N, device=dev), "edge_features": torch.randn(E, 300, device=dev)}, True, True)
    x28 = torch.nn.functional.one_hot(torch.from_numpy(b.atom_type), 28).float().to(dev)
    ef = torch.nn.functional.one_hot(torch.from_numpy(b.bond_type), 4).float().to(dev)
    kw = {"identifiers": (torch.rand(E, 12, device=dev) < 0.2).float(), "degrees": torch.zeros(N, device=dev), "edge_features": ef}
    gen = dict(d_in=28, d_ef=4, d_id=12, id_scope="local", d_msg=128, d_up=128, d_h=[128], msg_kind="general", **base)
    layer_case("GSN_edge_sparse general d=128 eval fwd (config 2 layer)", layers.GSN_edge_sparse(**gen), x28, kw, False, False)
    layer_case("GSN_edge_sparse general d=128 train fwd (batch-stat BN)", layers.GSN_edge_sparse(**gen), x28, kw, True, False)
    layer_case("GSN_edge_sparse general d=128 train fwd+bwd", layers.GSN_edge_sparse(**gen), x28, kw, True, True)
    # ---- batched preprocessing driver, dataset-level recoding, code-gather edge stage
    import time
    from collections import namedtuple
    from gsn_amd import dataset as gds, encoding, patterns
    res["next_rows"] = []
    Graph = namedtuple("Graph", ["node_features", "edge_mat", "edge_features", "label"])
    raw = []
    for i in range(12000):
        n, e = zb.graph(i)
        raw.append(Graph(torch.zeros(n, dtype=torch.long), torch.from_numpy(np.ascontiguousarray(e)), torch.ones(e.shape[1], dtype=torch.long), 0.0))
    dicts = []
    for el in cyc(range(3, 7)):
        sg, part, memb, aut = patterns.induced_edge_automorphism_orbits(edge_list=el, directed=False, directed_orbits=False)
        dicts.append({"subgraph": sg, "orbit_partition": part, "orbit_membership": memb, "aut_count": aut})
    params = {"induced": False, "directed": False}
    gds.prepare_graphs(raw[:64], dicts, params, True, "ZINC", "edge")
    t0 = time.perf_counter()
    prepared = gds.prepare_graphs(raw, dicts, params, True, "ZINC", "edge")
    dt = time.perf_counter() - t0
    res["next_rows"].append({"case": "prepare_graphs: ZINC-shape x12000, cycle 3..6 edge ids (host collate + 1 launch + per-graph split)",
                             "s": round(dt, 3), "graphs_per_s": round(12000 / dt, 1)})
    ids = [g.identifiers for g in prepared]
    encoding.one_hot_unique(ids[:8])
    t0 = time.perf_counter()
    enc = encoding.one_hot_unique(ids)
    dt = time.perf_counter() - t0
    res["next_rows"].append({"case": "one_hot_unique over %d rows x 4 columns (incl. host cat + H2D/D2H)" % enc.codes.shape[0],
                             "s": round(dt, 4), "d": enc.d})
    big = torch.randint(0, 40, (1 << 24, 4), device=dev)
    dt = timeit(lambda: encoding.column_range(big), reps=5, warm=2)
    res["next_rows"].append({"case": "gsn_column_range_hip 2^24 x 4 int64", "ms": round(dt * 1e3, 3), "GBps": round(big.numel() * 8 / dt / 1e9, 1)})

    bb = __import__("bench").make_batch(65536, 5)
    Nb, Eb = bb.num_nodes, bb.num_edges
    eib = torch.from_numpy(bb.edge_index).to(dev)
    xcodes = layers.Codes(torch.from_numpy(bb.atom_type).to(dev), [28])
    efcodes = layers.Codes(torch.from_numpy(bb.bond_type).to(dev), [4])
    idcodes = layers.Codes(torch.randint(0, 3, (Eb, 4), device=dev), [3, 3, 3, 3])
    torch.manual_seed(0)
    lay = layers.GSN_edge_sparse(**gen).to(dev).eval()
    flags.CODE_STATUS_CHECK = False
    degb = torch.zeros(Nb, device=dev)
    xd, idd, efd = xcodes.dense(), idcodes.dense(), efcodes.dense()
    for name, args_ in (("dense one-hot inputs (reference boundary)", (xd, idd, efd)), ("integer codes (weight-row gather edge stage)", (xcodes, idcodes, efcodes))):
        def run():
            with torch.no_grad():
                lay(args_[0], eib, identifiers=args_[1], degrees=degb, edge_features=args_[2])
        dt = timeit(run, reps=10, warm=3)
        flags.KERNEL_TIMER = {}
        run(); torch.cuda.synchronize()
        kt = {k: round(sum(a.elapsed_time(b_) for a, b_, _ in v), 3) for k, v in flags.KERNEL_TIMER.items()}
        flags.KERNEL_TIMER = None
        res["next_rows"].append({"case": "GSN_edge_sparse L0 general d=128 eval fwd, 65536 graphs, " + name, "ms": round(dt * 1e3, 3),
                                 "graphs_per_s": round(65536 / dt, 1), "kernels": kt})
    print(json.dumps(res))


if __name__ == "__main__":
    main()

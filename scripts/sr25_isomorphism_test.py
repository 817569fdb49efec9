#!/usr/bin/env python3
"""BASELINE configs[0] end to end on this package only (README.md:84 / :89 of the reference):

    raw graph6 file -> load_g6_graphs -> generate_dataset (batched HIP counting) -> encode(one_hot_unique)
    -> GNNSubstructures (random weights, 2 x 64, eval) -> pairwise distances of the graph embeddings -> failure rate

`--model GSN_sparse` (induced cycles k <= 6 as GSN-e identifiers) must separate all 105 pairs of the 15 strongly regular
SR(25,12,5,6) graphs; `--model MPNN_sparse` (no identifiers) separates none.  Usage:
    python scripts/sr25_isomorphism_test.py [--path tests/golden/raw] [--name sr251256] [--model GSN_sparse] [--k 6]"""
import argparse
import json
import os
import sys
import types

import networkx as nx
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gsn_amd import counting, dataset, encoding, models, patterns  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.abspath(__file__))
    ap.add_argument("--path", default=os.path.join(here, "..", "tests", "golden", "raw"))
    ap.add_argument("--name", default="sr251256")
    ap.add_argument("--model", default="GSN_sparse", choices=["GSN_sparse", "MPNN_sparse"])
    ap.add_argument("--k", type=int, default=6)
    ap.add_argument("--eps", type=float, default=1e-2)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    edge_lists = [list(nx.cycle_graph(k).edges) for k in range(3, args.k + 1)]
    graphs, n_cls, _, _, sizes = dataset.generate_dataset(
        args.path, args.name, args.k, counting.subgraph_counts2ids, counting.subgraph_isomorphism_edge_counts,
        patterns.induced_edge_automorphism_orbits, False, "cycle_graph", edge_list=edge_lists, induced=True, directed=False,
        directed_orbits=False)
    graphs, _, d_id, _, _ = encoding.encode(graphs, "one_hot_unique", None, ids={})
    # collate (PyG DataLoader semantics: node offsets, batch vector)
    off, xs, eis, ids, bat = 0, [], [], [], []
    for g, d in enumerate(graphs):
        xs.append(d.x); eis.append(d.edge_index + off); ids.append(d.identifiers)
        bat.append(torch.full((d.x.shape[0],), g, dtype=torch.long)); off += d.x.shape[0]
    data = types.SimpleNamespace(x=torch.cat(xs).to(dev), edge_index=torch.cat(eis, 1).to(dev), identifiers=torch.cat(ids).to(dev),
                                 batch=torch.cat(bat).to(dev), degrees=torch.zeros(off, device=dev))
    L, d = 2, 64
    kw = dict(seed=0, model_name=args.model, readout="sum", dropout_features=[0.0] * (L + 1), bn=[False] * L,
              final_projection=[False] * L + [True], inject_ids=False, inject_edge_features=False, random_features=False,
              id_scope="local", d_msg=[d] * L, d_out=[d] * L, d_h=[[d]] * L, aggr="add", flow="source_to_target",
              msg_kind="general", train_eps=[False] * L, activation_mlp="relu", bn_mlp=True, jk_mlp=True, degree_embedding="None",
              degree_as_tag=[False] * L, retain_features=[True] * L, multi_embedding_aggr="sum", input_node_encoder="None",
              d_out_node_encoder=d, edge_encoder="None", d_out_edge_encoder=[d] * L, id_embedding="one_hot_encoder",
              d_out_id_embedding=d, d_out_degree_embedding=d, extend_dims=True, activation="relu")
    torch.manual_seed(0)
    model = models.GNNSubstructures(1, 10, None, d_id, None, None, None, None, None, **kw).to(dev).eval()
    with torch.no_grad():
        emb = model(data)
    dist = torch.pdist(emb.double())
    fails = int((dist < args.eps).sum())
    print(json.dumps({"dataset": args.name, "graphs": len(graphs), "model": args.model, "k": args.k, "d_id": list(d_id),
                      "orbit_partition_sizes": sizes, "pairs": int(dist.numel()), "failures": fails,
                      "failure_rate": round(fails / max(int(dist.numel()), 1), 4), "min_distance": float(dist.min())}))


if __name__ == "__main__":
    main()

"""r06: what one large molecule costs a dataset's counting launch, and what grouping by adjacency width class (gsn_amd.dataset.prepare_graphs)
gets back: 41 127 molhiv-sized graphs (mean 25.5 vertices) + ONE 222-vertex graph, rings 3..6 in edge mode, counted (a) in one launch whose
instantiation follows the 222-vertex graph, (b) as two launches, one per class."""
import os
import sys
import time

import networkx as nx
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gsn_amd import synth  # noqa: E402
from gsn_amd.counting import CountPlan, count_batch  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
gs = [synth.zinc_shape_graph(rng, mean_n=25.5, sd_n=6.0) for _ in range(41127)]
n_big, ei_big = synth.er_graph(222, 251, seed=9)
plan = CountPlan.get([list(nx.cycle_graph(k).edges) for k in range(3, 7)], "edge", False)


def timed(graphs_lists, name):
    bs = []
    for gl in graphs_lists:
        b = synth.collate(gl)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        bs.append((t(b.node_ptr), t(b.edge_ptr), t(b.edge_index), int(np.diff(b.node_ptr).max()), int(np.diff(b.edge_ptr).max()),
                   torch.empty((b.num_edges, 4), dtype=torch.int64, device=dev)))

    def run():
        for npt, ept, ei, mn, me, out in bs:
            count_batch(plan, npt, ept, ei, ids_are_global=True, max_nodes=mn, max_edges=me, device=dev, check=False, out=out)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    n = sum(len(gl) for gl in graphs_lists)
    print("%-44s %8.3f ms   %10.0f graphs/s" % (name, dt * 1e3, n / dt), flush=True)
    return [b[5] for b in bs]


small = timed([gs], "41 127 molecules alone")
one = timed([gs + [(n_big, ei_big)]], "+ one 222-vertex graph, ONE launch")
two = timed([gs, [(n_big, ei_big)]], "+ one 222-vertex graph, one launch per class")
print("same identifiers:", bool(torch.equal(one[0], torch.cat(two))))

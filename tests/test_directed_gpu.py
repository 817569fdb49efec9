"""directed=True (main.py --directed) on a real MI355X: digraph patterns and targets, vertex counts, through the C ABI.

Golden: counts_directed.npz (the reference's automorphism_orbits / subgraph_isomorphism_vertex_counts with directed=True over
networkx's DiGraphMatcher); larger seeded digraphs against the oracle.  Integer work: bit-exact."""
import numpy as np
import pytest
import torch

from helpers import case_names, count_case, directed_patterns

pytestmark = pytest.mark.gpu


def _digraph(rng, n, m, loops=0, dups=0):
    src = rng.integers(0, n, size=m)
    dst = rng.integers(0, n, size=m)
    keep = src != dst
    ei = np.stack([src[keep], dst[keep]]).astype(np.int64)
    if dups:
        ei = np.concatenate([ei, ei[:, :dups]], axis=1)
    if loops:
        lv = rng.integers(0, n, size=loops)
        ei = np.concatenate([ei[:, :2], np.stack([lv, lv]), ei[:, 2:]], axis=1)
    return n, ei


@pytest.mark.parametrize("name", case_names("counts_directed"))
def test_golden_directed_counts(name):
    from gsn_amd.counting import CountPlan, count_batch
    c = count_case(name, "counts_directed")
    plan = CountPlan.get(c["patterns"], "vertex", c["induced"], False, directed=True)
    out, st = count_batch(plan, c["node_ptr"], c["edge_ptr"], c["edge_index_local"], ids_are_global=False)
    assert (st.cpu().numpy() == 0).all()
    assert np.array_equal(out.cpu().numpy(), c["counts"])


@pytest.mark.parametrize("induced", [False, True])
def test_directed_batch_vs_oracle(induced):
    """Digraphs of 20 .. 300 vertices (all four widths of the bit matrix that one wave handles, plus the 16-bit-id path), with
    repeated arcs and self loops; every golden pattern plus directed 5- and 6-cycles."""
    from gsn_amd import synth
    from gsn_amd.counting import counts2ids_batch
    from oracle import oracle
    rng = np.random.default_rng(17)
    graphs = [_digraph(rng, 20, 70), _digraph(rng, 64, 400, loops=4, dups=9), _digraph(rng, 65, 300), _digraph(rng, 128, 700, dups=20),
              _digraph(rng, 200, 900), _digraph(rng, 300, 1300, loops=2), (5, np.zeros((2, 0), np.int64))]
    graphs += [_digraph(rng, int(n), int(3 * n)) for n in rng.integers(8, 40, size=200)]
    pats = [el for el, _, _ in directed_patterns()]
    pats += [[(i, (i + 1) % 6) for i in range(6)], [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (0, 5)]]
    b = synth.collate(graphs)
    got = counts2ids_batch(b, pats, "vertex", induced, directed=True).cpu().numpy()
    local = b.edge_index - np.repeat(b.node_ptr[:-1], np.diff(b.edge_ptr))[None, :]
    ref = oracle.counts2ids("vertex", induced, b.node_ptr, b.edge_ptr, local, pats, n_threads=8, directed=True)
    assert np.array_equal(got, ref)
    assert got.sum() > 0


def test_symmetric_digraph_equals_undirected_counts():
    """A digraph with both arcs of every edge, matched with the bidirected version of an undirected pattern, has exactly the
    undirected pattern's occurrences (non-induced and induced)."""
    from gsn_amd import synth
    from gsn_amd.counting import counts2ids_batch
    b = synth.zinc_shape_batch(500, seed=4)
    und = [[(0, 1), (1, 2), (2, 0)], [(i, (i + 1) % 6) for i in range(6)], [(0, 1), (0, 2), (0, 3)], [(0, 1), (1, 2), (2, 3)]]
    both = [el + [(v, u) for u, v in el] for el in und]
    for induced in (False, True):
        a = counts2ids_batch(b, und, "vertex", induced)
        d = counts2ids_batch(b, both, "vertex", induced, directed=True)
        assert torch.equal(a, d) and int(a.sum()) > 0


def test_reference_signatures_directed():
    """automorphism_orbits(directed=True) -> subgraph_isomorphism_vertex_counts(directed=True) / subgraph_counts2ids with
    subgraph_params['directed'] = True, as utils_data_gen.py:31-42 / utils_ids.py:17-25 chain them."""
    from gsn_amd import patterns, counting
    name = "digraphs_mono_vertex"
    c = count_case(name, "counts_directed")
    dicts = []
    for el in c["patterns"]:
        sg, part, memb, aut = patterns.automorphism_orbits(edge_list=el, print_msgs=False, directed=True, directed_orbits=False)
        dicts.append({"subgraph": sg, "orbit_partition": part, "orbit_membership": memb, "aut_count": aut})
    g = 2                                                   # the graph with self loops and repeated arcs
    npt, ept = c["node_ptr"], c["edge_ptr"]
    ei = torch.from_numpy(c["edge_index_local"][:, ept[g]:ept[g + 1]])
    n = int(npt[g + 1] - npt[g])
    want = c["counts"][npt[g]:npt[g + 1]]
    col = 0
    for d in dicts:
        got = counting.subgraph_isomorphism_vertex_counts(ei, subgraph_dict=d, induced=False, num_nodes=n, directed=True)
        w = len(d["orbit_partition"])
        assert got.dtype == torch.float64 and np.array_equal(got.numpy(), want[:, col:col + w])
        col += w

    class Data:
        pass
    data = Data()
    data.x = torch.ones(n, 1)
    data.edge_index = ei
    res = counting.subgraph_counts2ids(counting.subgraph_isomorphism_vertex_counts, data, dicts, {"induced": False, "directed": True})
    assert np.array_equal(res.identifiers.numpy(), want)
    assert torch.equal(res.edge_index, ei[:, ei[0] != ei[1]])
    # the flag of the call and the flag the pattern was analysed under must agree
    with pytest.raises(ValueError):
        counting.subgraph_isomorphism_vertex_counts(ei, subgraph_dict=dicts[0], induced=False, num_nodes=n, directed=False)
    with pytest.raises(NotImplementedError):                 # the reference's directed edge counter: NameError
        counting.subgraph_counts2ids(counting.subgraph_isomorphism_edge_counts, data, dicts, {"induced": False, "directed": True})

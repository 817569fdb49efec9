"""The oracle (oracle/) against the golden vectors produced by the reference's own Python.
CPU only.  This is what pins the oracle; the GPU parity tests then compare the HIP path with the oracle."""
import numpy as np
import pytest
import torch

from helpers import load, case_names, count_case, layer_case, directed_patterns
from oracle import oracle


@pytest.mark.parametrize("name", case_names("orbits"))
def test_orbits(name):
    z = load("orbits")
    edges = z[name + "/edges"]
    memb, n_orb, aut = oracle.automorphism_orbits(edges)
    assert memb.tolist() == z[name + "/v_membership"].tolist()
    assert n_orb == int(z[name + "/n_vorbits"]) and aut == int(z[name + "/aut_count"])
    for sfx, d in (("", False), ("_dir", True)):
        arcs, am, ne, aut2 = oracle.induced_edge_orbits(edges, directed_orbits=d)
        assert arcs.tolist() == z[name + "/e_list" + sfx].tolist()
        assert am.tolist() == z[name + "/e_membership" + sfx].tolist()
        assert ne == int(z[name + "/n_eorbits" + sfx]) and aut2 == aut


@pytest.mark.parametrize("name", case_names("counts"))
def test_counts(name):
    c = count_case(name)
    got = oracle.counts2ids(c["mode"], c["induced"], c["node_ptr"], c["edge_ptr"], c["edge_index_local"], c["patterns"],
                            directed_orbits=c["directed_orbits"], n_threads=4)
    assert got.shape == c["counts"].shape
    assert np.array_equal(got, c["counts"])


@pytest.mark.parametrize("name", case_names("counts_cliques"))
def test_heavy_clique_counts(name):
    """K3..K5 on the IMDB-BINARY graphs the VF2-backed goldens leave out (> 1 M K5 maps each; the 136-vertex graph): per-vertex / per-edge
    tallies of networkx.enumerate_all_cliques -- an enumerator independent of VF2 and of this oracle (make_golden.py --only cliques)"""
    c = count_case(name, "counts_cliques")
    got = oracle.counts2ids(c["mode"], c["induced"], c["node_ptr"], c["edge_ptr"], c["edge_index_local"], c["patterns"],
                            directed_orbits=c["directed_orbits"], n_threads=4)
    assert np.array_equal(got, c["counts"])


@pytest.mark.parametrize("name", [n for n in case_names("counts_stars") if n.startswith("small_")])
def test_star8_counts(name):
    """star_graph(8), the 9-vertex pattern of --id_type star_graph --k 8 (utils.py:59-62): the reference's functions over the VF2 stand-in on
    graphs whose hubs have degree <= 9 (make_golden.py --only stars; the IMDB cases of that file come from a closed form pinned to these and
    are out of this all-maps enumerator's reach: 20! / 12! maps per hub), and the pattern's orbits."""
    c = count_case(name, "counts_stars")
    got = oracle.counts2ids(c["mode"], c["induced"], c["node_ptr"], c["edge_ptr"], c["edge_index_local"], c["patterns"], n_threads=4)
    assert np.array_equal(got, c["counts"])
    from helpers import load
    z = load("counts_stars")
    m, n_orb, a = oracle.automorphism_orbits(z["star8/edges"].tolist())
    assert m.tolist() == z["star8/v_membership"].tolist() and a == int(z["star8/aut_count"]) == 40320 and n_orb == 2


def test_directed_orbits_and_counts():
    """directed=True (main.py --directed): digraph patterns and targets, vertex counts (counts_directed.npz = the reference's
    automorphism_orbits / subgraph_isomorphism_vertex_counts with directed=True over networkx's DiGraphMatcher)."""
    for el, memb, aut in directed_patterns():
        m, n_orb, a = oracle.automorphism_orbits(el, directed=True)
        assert m.tolist() == memb and a == aut and n_orb == len(set(memb))
    for name in case_names("counts_directed"):
        c = count_case(name, "counts_directed")
        got = oracle.counts2ids("vertex", c["induced"], c["node_ptr"], c["edge_ptr"], c["edge_index_local"], c["patterns"], directed=True)
        assert np.array_equal(got, c["counts"]) and got.sum() > 0
    with pytest.raises(NotImplementedError):
        oracle.counts2ids("edge", False, [0, 3], [0, 2], np.array([[0, 1], [1, 2]]), [[(0, 1), (1, 2)]], directed=True)


def test_srg_closed_forms():
    """Independent of any VF2: in SR(25,12,5,6) every edge lies in lambda=5 triangles, every vertex in k*lambda/2=30."""
    c = count_case("sr25_cycle3-5_mono_vertex")
    assert (c["counts"][:, 0] == 30).all()
    c = count_case("sr25_cycle3-5_mono_edge")
    assert (c["counts"][:, 0] == 5).all()
    # SURVEY 8(a) anchors: C4 300/50, C5 3276/546 per vertex/edge, C6 5340 per edge
    assert (count_case("sr25_cycle3-5_mono_vertex")["counts"][:, 1] == 300).all()
    assert (count_case("sr25_cycle3-5_mono_edge")["counts"][:, 2] == 546).all()
    assert (count_case("sr25_cycle6_mono_edge_g0")["counts"][:, 0] == 5340).all()


def test_counts2ids_end_to_end():
    z = load("counts2ids")
    ptr, flat = z["pattern_ptr"], z["pattern_edges"]
    pats = [flat[ptr[i]:ptr[i + 1]].tolist() for i in range(len(ptr) - 1)]
    for mode in ("vertex", "edge"):
        ei = z[mode + "/in_edge_index"]
        ef = np.arange(ei.shape[1]) + 100
        ei2, ef2 = oracle.remove_self_loops(ei, ef)
        assert np.array_equal(ei2, z[mode + "/out_edge_index"]) and np.array_equal(ef2, z[mode + "/out_edge_features"])
        n = int(z[mode + "/in_num_nodes"])
        got = oracle.counts2ids(mode, False, [0, n], [0, ei2.shape[1]], ei2, pats)
        assert np.array_equal(got, z[mode + "/identifiers"])


def test_key_error_on_missing_direction():
    ei = np.array([[0, 1, 1, 2, 2], [1, 0, 2, 1, 0]], dtype=np.int64)  # (0,2) missing while (2,0) present; triangle exists
    with pytest.raises(KeyError):
        oracle.counts2ids("edge", False, [0, 3], [0, 5], ei, [[(0, 1), (1, 2), (2, 0)]])


@pytest.mark.parametrize("name", case_names("layers"))
def test_layers_forward(name):
    c = layer_case(name)
    sd = {k[3:]: torch.from_numpy(v) for k, v in c.items() if k.startswith("sd/")}
    t = lambda k: torch.from_numpy(c[k]) if k in c else None
    stats = {}
    y = oracle.layer_forward(c["cls"], c["ctor"], sd, t("x"), t("edge_index"), identifiers=t("identifiers"),
                             degrees=t("degrees"), edge_features=t("edge_features"), training=c["train"],
                             bn_stats_out=stats)
    ref = t("y")
    err = (y - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    assert err < 1e-5, err
    for k, v in stats.items():
        assert torch.allclose(v, t("sd_after/" + k), rtol=1e-5, atol=1e-6)

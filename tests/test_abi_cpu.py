"""CPU checks of the boundary: the C-ABI library loads and exports every symbol include/gsn_abi.h declares, the
host-side entry points (pattern orbits, plan compiler) agree with the golden vectors, and the product path refuses
to run without a GPU instead of silently falling back."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from helpers import load, case_names, count_case

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from gsn_amd import _abi
    _abi.build()
    return _abi.lib()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(REPO, "include", "gsn_abi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gsn_[a-z0-9_]+)\s*\(", hdr))
    assert {"gsn_count_hip", "gsn_pattern_orbits", "gsn_linear_fwd_hip", "gsn_propagate_fwd_hip"} <= declared
    raw = ctypes.CDLL(os.path.join(REPO, "gsn_amd", "lib", "libgsn_hip.so"))
    for name in declared:
        assert hasattr(raw, name), "libgsn_hip.so does not export %s" % name
    from gsn_amd import _abi
    assert declared == set(_abi.SIGNATURES), "ctypes signature table out of sync with include/gsn_abi.h"
    assert lib.gsn_version() == 1


@pytest.mark.parametrize("name", case_names("orbits"))
def test_pattern_orbits_match_reference(lib, name):
    from gsn_amd import patterns
    z = load("orbits")
    edges = z[name + "/edges"]
    k = int(edges.max()) + 1
    if k > 8:
        with pytest.raises(Exception):
            patterns.analyse(edges)
        return
    info = patterns.analyse(edges, False)
    assert info["vertex_orbit"].tolist() == z[name + "/v_membership"].tolist()
    assert info["aut_count"] == int(z[name + "/aut_count"])
    assert info["arcs"].tolist() == z[name + "/e_list"].tolist()
    assert info["arc_orbit"].tolist() == z[name + "/e_membership"].tolist()
    info = patterns.analyse(edges, True)
    assert info["arc_orbit"].tolist() == z[name + "/e_membership_dir"].tolist()
    assert info["n_edge_orbits"] == int(z[name + "/n_eorbits_dir"])


def test_reference_style_return_values(lib, capsys):
    from gsn_amd import patterns
    g, part, memb, aut = patterns.automorphism_orbits(edge_list=[(0, 1), (1, 2), (2, 3)], directed=False, directed_orbits=False)
    assert part == {0: [0, 3], 1: [1, 2]} and memb == {0: 0, 1: 1, 2: 1, 3: 0} and aut == 2
    assert "Automorphism count: 2" in capsys.readouterr().out
    g, epart, ememb, aut = patterns.induced_edge_automorphism_orbits(edge_list=[(0, 1), (1, 2), (2, 3)], directed=False, directed_orbits=False)
    assert epart == {0: [(0, 1), (1, 0), (2, 3), (3, 2)], 1: [(1, 2), (2, 1)]}
    assert ememb == {0: 0, 1: 0, 2: 1, 3: 1, 4: 0, 5: 0}
    import pickle
    assert pickle.loads(pickle.dumps(g)).edge_list == g.edge_list   # joblib workers pickle subgraph_dicts
    g, lpart, lmemb, aut = patterns.edge_automorphism_orbits(edge_list=[(0, 1), (1, 2), (2, 3)])
    assert lpart == {0: [(0, 1), (2, 3)], 1: [(1, 2)]} and lmemb == {0: 0, 1: 1, 2: 0} and aut == 2
    with pytest.raises(KeyError):       # single edge: its line graph has no edges, hence no graph-tool vertices (:211-212)
        patterns.edge_automorphism_orbits(edge_list=[(0, 1)])


def _line_cases():
    z = load("orbits")
    return [str(n) for n in z["names"] if (str(n) + "/line_membership") in z.files]


@pytest.mark.parametrize("name", _line_cases())
def test_line_graph_edge_orbits_match_reference(lib, name, capsys):
    """a3: the deprecated --edge_automorphism line_graph variant (utils_graph_processing.py:189-251) against what the
    reference's own function returned for 62 patterns (incl. K4 / diamond, where Aut(L(H)) is larger than Aut(H))."""
    from gsn_amd import patterns
    z = load("orbits")
    edges = [tuple(e) for e in z[name + "/edges"].tolist()]
    g, part, memb, aut = patterns.edge_automorphism_orbits(edge_list=edges, directed=False)
    assert [memb[i] for i in range(len(memb))] == z[name + "/line_membership"].tolist()
    assert len(part) == int(z[name + "/line_n_orbits"])
    assert [(o, e[0], e[1]) for o in sorted(part) for e in part[o]] == [tuple(r) for r in z[name + "/line_partition"].tolist()]
    assert aut == int(z[name + "/aut_count"])
    assert "Number of edge orbits: %d" % len(part) in capsys.readouterr().out


def test_graph_vertex_orbits_generic(lib):
    from gsn_amd import patterns
    import networkx as nx
    orb, n = patterns.graph_vertex_orbits(10, list(nx.petersen_graph().edges))          # vertex-transitive
    assert n == 1 and orb.tolist() == [0] * 10
    orb, n = patterns.graph_vertex_orbits(28, list(nx.relabel_nodes(nx.line_graph(nx.complete_graph(8)), {e: i for i, e in enumerate(nx.line_graph(nx.complete_graph(8)).nodes)}).edges))
    assert n == 1                                                                        # L(K8): 28 vertices, one orbit
    orb, n = patterns.graph_vertex_orbits(5, [(0, 1), (1, 2), (2, 3), (3, 4)])
    assert orb.tolist() == [0, 1, 2, 1, 0] and n == 3
    orb, n = patterns.graph_vertex_orbits(4, [(0, 1)])                                   # isolated vertices form an orbit
    assert orb.tolist() == [0, 0, 1, 1]
    with pytest.raises(Exception):
        patterns.graph_vertex_orbits(65, [(0, 1)])


@pytest.fixture(scope="module")
def harness():
    so = os.path.join(REPO, "tests", "_build", "libharness.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    srcs = [os.path.join(REPO, "tests", "host_harness.cpp"), os.path.join(REPO, "gsn_amd", "csrc", "patterns.cpp")]
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so] + srcs)
    return ctypes.CDLL(so)


@pytest.mark.parametrize("name", case_names("counts"))
def test_plans_and_search_core_on_host(lib, harness, name):
    """Plan compiler (product code) + per-lane search core (shared with the HIP kernel), run on the host by the
    test-only harness, against the golden counts of the reference."""
    from gsn_amd.counting import CountPlan
    c = count_case(name)
    plan = CountPlan(c["patterns"], c["mode"], c["induced"], c["directed_orbits"])
    npt, ept, ei = c["node_ptr"], c["edge_ptr"], c["edge_index_local"]
    I64P, U32P = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_uint32)
    outs = []
    for g in range(len(npt) - 1):
        n, E = int(npt[g + 1] - npt[g]), int(ept[g + 1] - ept[g])
        src = np.ascontiguousarray(ei[0, ept[g]:ept[g + 1]])
        dst = np.ascontiguousarray(ei[1, ept[g]:ept[g + 1]])
        rows = E if c["mode"] == "edge" else n
        out = np.zeros((rows, plan.n_cols), dtype=np.int64)
        st = harness.harness_count(plan.table.ctypes.data_as(U32P), ctypes.c_int64(n), ctypes.c_int64(E),
                                   src.ctypes.data_as(I64P), dst.ctypes.data_as(I64P), out.ctypes.data_as(I64P))
        assert st == 0
        outs.append(out)
    got = np.concatenate(outs, axis=0) if outs else np.zeros((0, plan.n_cols), np.int64)
    assert np.array_equal(got, c["counts"])


def test_directed_patterns_orbits_and_search_core_on_host(lib, harness):
    """directed=True: gsn_pattern_orbits with GSN_FLAG_DIRECTED against the reference's orbits, and the directed plans + the
    DIR search core on the host against the reference's directed vertex counts."""
    from gsn_amd import patterns
    from gsn_amd.counting import CountPlan
    from helpers import directed_patterns
    for el, memb, aut in directed_patterns():
        g, part, om, a = patterns.automorphism_orbits(edge_list=el, print_msgs=False, directed=True)
        assert [om[v] for v in range(len(memb))] == memb and a == aut and g.directed
        assert {o: sorted(vs) for o, vs in part.items()} == {o: [v for v in range(len(memb)) if memb[v] == o] for o in set(memb)}
    I64P, U32P = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_uint32)
    for name in case_names("counts_directed"):
        c = count_case(name, "counts_directed")
        plan = CountPlan(c["patterns"], "vertex", c["induced"], False, directed=True)
        assert int(plan.table[6]) == 2
        npt, ept, ei = c["node_ptr"], c["edge_ptr"], c["edge_index_local"]
        outs = []
        for g in range(len(npt) - 1):
            n, E = int(npt[g + 1] - npt[g]), int(ept[g + 1] - ept[g])
            src = np.ascontiguousarray(ei[0, ept[g]:ept[g + 1]])
            dst = np.ascontiguousarray(ei[1, ept[g]:ept[g + 1]])
            out = np.zeros((n, plan.n_cols), dtype=np.int64)
            st = harness.harness_count(plan.table.ctypes.data_as(U32P), ctypes.c_int64(n), ctypes.c_int64(E),
                                       src.ctypes.data_as(I64P), dst.ctypes.data_as(I64P), out.ctypes.data_as(I64P))
            assert st == 0
            outs.append(out)
        assert np.array_equal(np.concatenate(outs, axis=0), c["counts"])
    with pytest.raises(RuntimeError):                      # the reference's directed edge counter is broken: refused
        CountPlan([[(0, 1), (1, 2)]], "edge", False, False, directed=True)
    # an undirected pattern handed to a directed count (or the reverse) is refused, not silently mis-counted
    from gsn_amd import counting
    g_und = patterns.PatternGraph([(0, 1), (1, 2)])
    with pytest.raises(ValueError):
        counting._directed_of([g_und], True)


def test_plan_table_column_order(lib):
    """The plan table carries the order in which the kernel hands out a graph's (column, row) cells: a permutation of the columns,
    larger cycles before smaller ones (estimated search cost), tree-like five-vertex patterns before dense ones."""
    import networkx as nx
    from gsn_amd.counting import CountPlan
    plan = CountPlan([list(nx.cycle_graph(k).edges) for k in range(3, 9)], "edge", False)
    n = plan.n_cols
    order = plan.table[8 + n + 1: 8 + 2 * n + 1].tolist()
    assert sorted(order) == list(range(n)) and order == list(range(n - 1, -1, -1))
    assert int(plan.table[7]) == 8 + 2 * n + 1
    pats = [list(nx.path_graph(5).edges), list(nx.complete_graph(5).edges), list(nx.star_graph(4).edges)]
    plan = CountPlan(pats, "vertex", False)
    order = plan.table[8 + plan.n_cols + 1: 8 + 2 * plan.n_cols + 1].tolist()
    assert sorted(order) == list(range(plan.n_cols))
    k5_col = plan.n_cols - 3          # columns: path (3 orbits), K5 (1), star (2)
    assert order.index(3) > order.index(0), order      # the clique's column comes after the path's
    del k5_col


def test_no_cpu_fallback(lib):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gsn_amd import counting, layers, patterns
    _, part, memb, aut = patterns.automorphism_orbits(edge_list=[(0, 1), (1, 2), (2, 0)], print_msgs=False)
    d = {"subgraph": patterns.PatternGraph([(0, 1), (1, 2), (2, 0)]), "orbit_partition": part, "orbit_membership": memb, "aut_count": aut}
    ei = torch.tensor([[0, 1, 1, 2, 2, 0], [1, 0, 2, 1, 0, 2]])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        counting.subgraph_isomorphism_vertex_counts(ei, subgraph_dict=d, induced=False, num_nodes=3)
    layer = layers.MPNN_sparse(d_in=4, d_degree=1, degree_as_tag=False, retain_features=True, d_msg=4, d_up=4, d_h=[4], seed=0,
                               activation_name="relu", bn=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(torch.randn(3, 4), ei, degrees=torch.zeros(3))


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "gsn_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def test_state_dict_keys_and_seeded_init_match_reference():
    """Same constructor + same torch seed -> same parameter tensors as the reference layer (golden 'sd/')."""
    from gsn_amd import layers
    from helpers import layer_case
    for name in ["GSN_edge_sparse/zinc/general/local/source_to_target/bn1/train0",
                 "GSN_edge_sparse/zinc/gin/local/source_to_target/bn1/train0",
                 "GSN_sparse/zinc/gin/global/source_to_target/bn1/train0",
                 "GSN_edge_sparse_ogb/zinc/local/train0", "MPNN_sparse/zinc/general/source_to_target/train0"]:
        c = layer_case(name)
        layer = getattr(layers, c["cls"])(**c["ctor"])
        keys = {k[3:] for k in c if k.startswith("sd/")}
        assert set(layer.state_dict().keys()) == keys, name
        layer.load_state_dict({k: torch.from_numpy(c["sd/" + k]) for k in keys})


@pytest.mark.parametrize("n,mode", [(300, "vertex"), (620, "edge"), (768, "vertex")])
def test_search_core_with_16_bit_vertex_ids_on_host(lib, harness, n, mode):
    """Graphs beyond 256 vertices pack the partial map 16 bits per level (W = 8 / 12): plan + search core on the host vs
    the oracle."""
    import networkx as nx
    from gsn_amd import synth
    from gsn_amd.counting import CountPlan
    from oracle import oracle
    rng = np.random.default_rng(n)
    nn, ei = synth.zinc_shape_graph(rng, mean_n=n, sd_n=0.0, n_min=n, n_max=n, ring_rate=n / 12.0)
    pats = [list(nx.cycle_graph(k).edges) for k in (3, 4, 5, 6)] + [list(nx.star_graph(3).edges)]
    plan = CountPlan(pats, mode, False, False)
    E = ei.shape[1]
    I64P, U32P = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_uint32)
    src, dst = np.ascontiguousarray(ei[0]), np.ascontiguousarray(ei[1])
    out = np.zeros((E if mode == "edge" else nn, plan.n_cols), dtype=np.int64)
    st = harness.harness_count(plan.table.ctypes.data_as(U32P), ctypes.c_int64(nn), ctypes.c_int64(E),
                               src.ctypes.data_as(I64P), dst.ctypes.data_as(I64P), out.ctypes.data_as(I64P))
    assert st == 0
    ref = oracle.counts2ids(mode, False, np.array([0, nn]), np.array([0, E]), ei, pats, n_threads=4)
    assert np.array_equal(out, ref) and out.sum() > 0


def test_abi_header_is_plain_c():
    """include/gsn_abi.h is the drop-in boundary: it must compile as C99 and as C++ on its own (no torch / HIP types)."""
    hdr = os.path.join(REPO, "include", "gsn_abi.h")
    subprocess.check_call(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", hdr])
    subprocess.check_call(["g++", "-fsyntax-only", "-x", "c++", "-std=c++17", "-Wall", "-Werror", hdr])
    import re
    code = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)      # declarations only (comments cite torch calls)
    assert "torch" not in code.lower() and "hipStream_t" not in code and "#include <hip" not in code


def test_register_resident_layer_kernels_build_without_spills(tmp_path):
    """csrc/layer_rr.hip and csrc/layer_w.hip run one or two waves per SIMD that wait only for their own loads: a spilled register's
    reload is a `s_waitcnt vmcnt(0)` behind every gather in flight (measured on layer_w.hip: 4 000-10 000 cycles per tile, 6-7 % of
    the layer).  Whether the allocator spills flips with small edits, so the product kernels are checked here: device-only assembly
    from hipcc (cross-compiles without a GPU), `.vgpr_spill_count` of the non-diagnostic instantiations must not grow (0 for the one-wave-per-SIMD kernel)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gsn_amd", "csrc")
    # (layer_rr.hip, two waves per SIMD: 10 registers that live across the tile loop are spilled; their reloads sit behind the edge stage)
    for name, kernel, allowed in (("layer_w.hip", "layer_fused_kernel_wILb0E", 0), ("layer_rr.hip", "layer_fused_kernel_rrILi4ELi2ELb0E", 10)):
        out = tmp_path / (name + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", os.path.join(src, name), "-o", str(out)],
                       check=True, capture_output=True, timeout=600)
        text = out.read_text()
        blocks = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)
        hits = [(n, int(v)) for n, v in blocks if kernel in n]
        assert hits, name
        assert all(v <= allowed for _, v in hits), hits


@pytest.mark.parametrize("mode", ["vertex", "edge"])
@pytest.mark.parametrize("induced", [False, True])
def test_closed_form_tails_of_the_search_core_on_host(lib, harness, mode, induced):
    """count_core.h's closed forms of the last two levels (independent pendants, twins, chain tails through degree bit planes) and its tight
    tail loop, run on the host by the test-only harness against the oracle: patterns with pendant vertices on molecule-shaped graphs and on
    Erdos-Renyi graphs of 70 / 130 vertices (one / two / four-word bit rows).  (The same patterns on the GPU: test_count_gpu.py.)"""
    from gsn_amd import synth
    from gsn_amd.counting import CountPlan
    from oracle import oracle
    pats = [[(0, 1), (0, 2), (0, 3), (0, 4)], [(0, 1), (1, 2), (2, 3), (3, 4)], [(0, 1), (1, 2), (2, 3), (2, 4)],
            [(0, 1), (1, 2), (2, 0), (2, 3), (3, 4)], [(0, 1), (1, 2), (2, 3), (3, 0), (0, 4)], [(0, 1), (1, 2), (2, 0), (0, 3), (1, 4)],
            [(0, 1), (0, 2), (0, 3), (1, 4), (2, 5)],
            [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5)], [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6)],      # stars: runs of 5 / 4 .. twin leaves
            [(0, 1), (1, 2), (2, 3), (2, 4), (2, 5), (2, 6)]]                                                # a broom: path + four twin leaves
    from gsn_amd.counting import PLAN_STRIDE_WORDS as PS
    plan = CountPlan(pats, mode, induced, False)
    n_plans, plans_off = int(plan.table[3]), int(plan.table[7])
    kinds = {(int(plan.table[plans_off + i * PS + 1]) >> 28) & 3 for i in range(n_plans)}
    assert kinds == ({0} if induced else {0, 1, 2, 3}), kinds
    runs = {2 + (int(plan.table[plans_off + i * PS + 1]) >> 30) for i in range(n_plans) if (int(plan.table[plans_off + i * PS + 1]) >> 28) & 3 == 2}
    assert induced or {2, 3, 4, 5} <= runs, runs                 # C(n, r) for r = 2 .. 5 twin levels
    b0 = synth.zinc_shape_batch(6, seed=3)
    graphs = [(int(b0.node_ptr[g + 1] - b0.node_ptr[g]), b0.edge_index[:, b0.edge_ptr[g]:b0.edge_ptr[g + 1]] - b0.node_ptr[g]) for g in range(6)]
    for n_, m_, s_ in ((70, 160, 1), (130, 300, 2), (40, 200, 3)):
        n_g, ei_g = synth.er_graph(n_, m_, s_)
        graphs.append((n_g, np.asarray(ei_g)))
    I64P, U32P = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_uint32)
    for n, ei in graphs:
        ei = np.ascontiguousarray(ei.astype(np.int64))
        E = ei.shape[1]
        src, dst = np.ascontiguousarray(ei[0]), np.ascontiguousarray(ei[1])
        out = np.zeros((E if mode == "edge" else n, plan.n_cols), dtype=np.int64)
        st = harness.harness_count(plan.table.ctypes.data_as(U32P), ctypes.c_int64(n), ctypes.c_int64(E),
                                   src.ctypes.data_as(I64P), dst.ctypes.data_as(I64P), out.ctypes.data_as(I64P))
        assert st == 0
        ref = oracle.counts2ids(mode, induced, np.array([0, n], dtype=np.int64), np.array([0, E], dtype=np.int64), ei, pats, n_threads=4)
        assert np.array_equal(out, ref), (mode, induced, n)

#!/usr/bin/env python3
"""BUILD-CONTAINER check (needs /root/reference; never runs on the GPU box): the reference's OWN caller code binds to the drop-ins.

`gsn_amd/dropin` is put in front of /root/reference on sys.path -- the integration route of INTEGRATION.md 1 -- and the reference's
unchanged `utils.py` is imported from /root/reference.  Its import statements (utils.py:2-5) then resolve utils_graph_processing /
utils_ids / utils_data_gen / utils_graph_learning to this package, and its host-only logic runs over them:

  * `process_arguments` (utils.py:35-92) picks `count_fn`, `automorphism_fn`, `extract_id_fn` for id_scope local / global and
    edge_automorphism induced / line_graph: asserted to BE this package's functions, with the `__name__` strings utils_data_gen.py:103
    compares against;
  * `get_custom_edge_list` (utils.py:16-33) builds the pattern families (cycle_graph, complete_graph, path_graph, star_graph,
    all_simple_graphs from the reference's .g6 files, ...);
  * the `subgraph_dicts` construction of utils_data_gen.py:35-42 (the same four-tuple unpacking, `len(orbit_partition)`) is run over the
    selected `automorphism_fn` for every pattern, and the tables are compared with tests/golden/orbits.npz (produced by the reference's
    own functions over networkx VF2).

No device is touched (orbits are host code: csrc/patterns.cpp).  Exit code 0 and "binding ok" on success.
    python scripts/check_reference_binding.py"""
import os
import sys
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (build container only)")
    sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
    import make_golden                                   # (its stubs of torch_geometric / ogb / graph_tool: what utils.py imports but this check never calls)
    make_golden.install_stubs()
    sys.modules.pop("graph_tool", None)                  # the drop-ins must not need graph-tool at all
    sys.path[:0] = [os.path.join(REPO, "gsn_amd", "dropin"), REPO]
    sys.path.append(REF)                                 # BEHIND the drop-ins: what they do not shadow (utils.py, utils_misc.py, ...) comes from the reference
    import utils                                         # the reference's utils.py, unchanged
    assert utils.__file__ == os.path.join(REF, "utils.py"), utils.__file__
    import gsn_amd.counting as gc
    import gsn_amd.dataset as gd
    import gsn_amd.patterns as gp
    import utils_graph_processing as ugp
    assert ugp.__file__.startswith(os.path.join(REPO, "gsn_amd", "dropin")), ugp.__file__
    assert utils.subgraph_counts2ids is gc.subgraph_counts2ids
    assert utils.generate_dataset.__wrapped__ is gd.generate_dataset

    base = dict(inject_degrees=False, degree_as_tag=False, retain_features=True, num_layers=4, d_msg=None, d_out=64, d_h=None,
                num_mlp_layers=2, d_out_edge_encoder=None, d_out_node_encoder=None, d_out_id_embedding=None, d_out_degree_embedding=None,
                root_folder=os.path.join(REF, "datasets"), custom_edge_list=None, vn=False, train_eps=False, final_projection=[True], bn=True,
                dropout_features=0.0, loss_fn="L1Loss", prediction_fn="L1Loss", regression=True)
    want = {("local", "induced"): (gc.subgraph_isomorphism_edge_counts, gp.induced_edge_automorphism_orbits),
            ("global", "induced"): (gc.subgraph_isomorphism_vertex_counts, gp.automorphism_orbits),
            ("local", "line_graph"): (gc.subgraph_isomorphism_edge_counts, gp.edge_automorphism_orbits),
            ("global", "line_graph"): (gc.subgraph_isomorphism_vertex_counts, gp.automorphism_orbits)}
    z = np.load(os.path.join(REPO, "tests", "golden", "orbits.npz"), allow_pickle=False)
    families = [("cycle_graph", [8], "cycle_graph", 0), ("complete_graph", [6], "complete_graph", 0), ("path_graph", [6], "path_graph", 0),
                ("star_graph", [5], "star_graph", 0), ("diamond_graph", [4], "diamond_graph", 0),
                ("all_simple_graphs", [5], None, 0), ("all_simple_graphs_chosen_k", [5], "all_simple_graphs_5", 0),
                ("cycle_graph_chosen_k", [4, 6], "cycle_graph", None)]
    n_patterns = 0
    for (scope, autom), (count_want, aut_want) in want.items():
        for id_type, k, fixture, _ in families:
            args = dict(base, id_scope=scope, edge_automorphism=autom, id_type=id_type, k=list(k))
            res = utils.process_arguments(args)
            args_out, extract_id_fn, count_fn, automorphism_fn = res[0], res[1], res[2], res[3]
            assert extract_id_fn is gc.subgraph_counts2ids
            assert count_fn is count_want, (scope, autom, count_fn)
            assert automorphism_fn is aut_want, (scope, autom, automorphism_fn)
            # the strings utils_data_gen.py:103 compares
            assert count_fn.__name__ == ("subgraph_isomorphism_edge_counts" if scope == "local" else "subgraph_isomorphism_vertex_counts")
            edge_lists = args_out["custom_edge_list"]
            if autom == "line_graph" and scope == "local":
                continue                                  # (deprecated path: covered by test_line_graph_edge_orbits_match_reference)
            # utils_data_gen.py:35-42, statement for statement over the selected function
            subgraph_dicts, orbit_partition_sizes = [], []
            for edge_list in edge_lists:
                subgraph, orbit_partition, orbit_membership, aut_count = automorphism_fn(edge_list=edge_list, directed=False, directed_orbits=False)
                subgraph_dicts.append({'subgraph': subgraph, 'orbit_partition': orbit_partition, 'orbit_membership': orbit_membership, 'aut_count': aut_count})
                orbit_partition_sizes.append(len(orbit_partition))
            # against the reference-generated tables
            if id_type == "all_simple_graphs":
                keys = ["all_simple_graphs_%d/%d" % (kk, i) for kk in (3, 4, 5) for i in range(sum(1 for f in z.files if f.startswith("all_simple_graphs_%d/" % kk) and f.endswith("/edges")))]
            elif id_type == "cycle_graph_chosen_k":
                keys = ["cycle_graph/%d" % (kk - 3) for kk in k]
            else:
                keys = ["%s/%d" % (fixture, i) for i in range(len(edge_lists))]
            assert len(keys) == len(edge_lists), (id_type, len(keys), len(edge_lists))
            for key, el, d in zip(keys, edge_lists, subgraph_dicts):
                assert [list(e) for e in el] == z[key + "/edges"].tolist(), key
                assert int(d['aut_count']) == int(z[key + "/aut_count"]), key
                memb = d['orbit_membership']
                if scope == "global":
                    assert [int(memb[v]) for v in sorted(memb)] == z[key + "/v_membership"].tolist(), key
                    assert len(d['orbit_partition']) == int(z[key + "/n_vorbits"]), key
                else:
                    arcs = z[key + "/e_list"].tolist()
                    # (the reference keys the edge membership by the index of the arc in its sorted arc list, utils_graph_processing.py:84-98)
                    assert [int(memb[i]) for i in range(len(arcs))] == z[key + "/e_membership"].tolist(), key
                    assert sorted(tuple(a) for p_ in d['orbit_partition'].values() for a in p_) == sorted(tuple(a) for a in arcs), key
                    assert len(d['orbit_partition']) == int(z[key + "/n_eorbits"]), key
                n_patterns += 1
    # --id_type star_graph --k 8 (utils.py:59-62): its largest pattern has NINE vertices (GSN_KMAX 8 -> 9 in r06); tables of counts_stars.npz
    zs = np.load(os.path.join(REPO, "tests", "golden", "counts_stars.npz"), allow_pickle=False)
    for scope in ("global", "local"):
        args = dict(base, id_scope=scope, edge_automorphism="induced", id_type="star_graph", k=[8])
        res = utils.process_arguments(args)
        el = res[0]["custom_edge_list"][-1]
        assert [list(e) for e in el] == zs["star8/edges"].tolist()
        subgraph, orbit_partition, orbit_membership, aut_count = res[3](edge_list=el, directed=False, directed_orbits=False)
        assert int(aut_count) == int(zs["star8/aut_count"]) == 40320
        key = "star8/v_membership" if scope == "global" else "star8/e_membership"
        assert [int(orbit_membership[i]) for i in range(len(orbit_membership))] == zs[key].tolist()
        n_patterns += 1
    print("binding ok: reference utils.process_arguments / get_custom_edge_list over gsn_amd/dropin, %d pattern tables equal to orbits.npz" % n_patterns)


if __name__ == "__main__":
    main()

"""The drop-in directory resolves the reference's import statements (utils.py:2-5, main.py:19, utils_data_gen.py:6,
models_graph_classification.py:5-8, models_graph_classification_ogb_original.py:7-8) to our implementations."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r'''
import sys
from utils_graph_processing import subgraph_isomorphism_edge_counts, subgraph_isomorphism_vertex_counts, \
    induced_edge_automorphism_orbits, edge_automorphism_orbits, automorphism_orbits
from utils_ids import subgraph_counts2ids
from graph_filters.GSN_sparse import GSN_sparse
from graph_filters.GSN_edge_sparse import GSN_edge_sparse
from graph_filters.MPNN_sparse import MPNN_sparse
from graph_filters.MPNN_edge_sparse import MPNN_edge_sparse
from graph_filters.GSN_edge_sparse_ogb import GSN_edge_sparse_ogb
from graph_filters.MPNN_edge_sparse_ogb import MPNN_edge_sparse_ogb
from utils_data_gen import generate_dataset
from utils_encoding import encode
from utils_graph_learning import multi_class_accuracy, global_add_pool_sparse, global_mean_pool_sparse, \
    DiscreteEmbedding, central_encoder
from models_misc import mlp, choose_activation
import gsn_amd.layers, gsn_amd.counting, gsn_amd.encoding, gsn_amd.dataset
assert encode is gsn_amd.encoding.encode and mlp is gsn_amd.layers.mlp
assert generate_dataset.__wrapped__ is gsn_amd.dataset.generate_dataset
import torch
assert float(multi_class_accuracy(torch.tensor([[0.1, 0.9], [0.8, 0.2]]), torch.tensor([1, 1]))) == 1.0
assert GSN_edge_sparse is gsn_amd.layers.GSN_edge_sparse
assert subgraph_counts2ids is gsn_amd.counting.subgraph_counts2ids
# how the reference picks the functions (utils.py:40-48) and tells the modes apart (utils_data_gen.py:103)
count_fn = subgraph_isomorphism_edge_counts
assert count_fn.__name__ == 'subgraph_isomorphism_edge_counts'
assert subgraph_isomorphism_vertex_counts.__name__ == 'subgraph_isomorphism_vertex_counts'
sg, part, memb, aut = induced_edge_automorphism_orbits(edge_list=[(0, 1), (1, 2), (2, 0)], directed=False, directed_orbits=False)
assert len(part) == 1 and aut == 6
print("dropin ok")
'''


def test_dropin_imports_resolve_to_gsn_amd():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(REPO, "gsn_amd", "dropin"), REPO])
    out = subprocess.run([sys.executable, "-c", SNIPPET], env=env, cwd="/tmp", capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "dropin ok" in out.stdout


def test_reference_callers_bind_to_the_dropins():
    """scripts/check_reference_binding.py: the reference's unchanged utils.py (process_arguments, get_custom_edge_list) and the
    subgraph_dicts loop of utils_data_gen.py:35-42 over gsn_amd/dropin -- selected callables are ours, 213 orbit tables equal orbits.npz.
    Build container only (/root/reference does not exist on the GPU box)."""
    import pytest
    if not os.path.isdir("/root/reference"):
        pytest.skip("needs /root/reference")
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "check_reference_binding.py")], env=env, cwd="/tmp",
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "binding ok" in out.stdout

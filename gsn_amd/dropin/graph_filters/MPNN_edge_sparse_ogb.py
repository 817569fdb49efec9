"""Drop-in for the reference's graph_filters/MPNN_edge_sparse_ogb.py: same import path and class, HIP kernels underneath."""
from gsn_amd.layers import MPNN_edge_sparse_ogb  # noqa: F401

"""Drop-in for the reference's graph_filters/MPNN_sparse.py: same import path and class, HIP kernels underneath."""
from gsn_amd.layers import MPNN_sparse  # noqa: F401

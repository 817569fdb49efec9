"""Drop-in for the reference's graph_filters/GSN_sparse.py: same import path and class, HIP kernels underneath."""
from gsn_amd.layers import GSN_sparse  # noqa: F401

"""Drop-in for the reference's models_misc.py (imported by graph_filters/* and models_graph_classification*.py:10)."""
from gsn_amd.layers import mlp, choose_activation  # noqa: F401

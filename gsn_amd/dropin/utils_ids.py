"""Drop-in for the reference's utils_ids.py (imported by utils.py:3)."""
from gsn_amd.counting import subgraph_counts2ids  # noqa: F401

"""Drop-in for the reference's utils_graph_processing.py (imported by utils.py:2 and utils_data_gen.py:6):
same function names and signatures, computed by libgsn_hip.so (no graph-tool)."""
from gsn_amd.patterns import (automorphism_orbits, induced_edge_automorphism_orbits,  # noqa: F401
                              edge_automorphism_orbits)
from gsn_amd.counting import (subgraph_isomorphism_vertex_counts,  # noqa: F401
                              subgraph_isomorphism_edge_counts)

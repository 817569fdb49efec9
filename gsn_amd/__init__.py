"""gsn_amd -- MI355X-native (gfx950) hot paths of Graph Substructure Networks.

HP-1  substructure / orbit counting        gsn_amd.patterns, gsn_amd.counting      (gsn_count_hip)
HP-2  sparse GSN message-passing layers    gsn_amd.layers                          (gsn_linear_fwd_hip, gsn_propagate_*_hip)

``gsn_amd/dropin`` holds modules with the reference's own import paths (utils_graph_processing, utils_ids,
graph_filters.*): put it in front of the reference on PYTHONPATH and ``main.py`` runs on these kernels unchanged
(INTEGRATION.md).  The C ABI is declared in include/gsn_abi.h.  There is no CPU fallback.
"""
__version__ = "0.1.0"

"""ctypes binding of libgsn_hip.so (the C ABI declared in include/gsn_abi.h).

There is no CPU fallback: if the shared library is missing, or a device entry point is called without a
visible gfx950 GPU, this module raises.  ``build()`` compiles the library in-tree with hipcc
(``gsn_amd/csrc/Makefile`` -> ``gsn_amd/lib/libgsn_hip.so``); hipcc cross-compiles gfx950 without a GPU.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSN_LIB_PATH") or os.path.join(_HERE, "lib", "libgsn_hip.so")   # (GSN_LIB_PATH: A/B builds of the same ABI)
_lib = None

c_i64 = ctypes.c_int64
c_i32 = ctypes.c_int32
c_int = ctypes.c_int
c_vp = ctypes.c_void_p


class GsnError(RuntimeError):
    """A libgsn_hip.so entry point returned a negative status."""


class gsn_block(ctypes.Structure):
    _fields_ = [("data", c_vp), ("idx", c_vp), ("width", c_i64), ("idx32", c_vp)]


class gsn_chain_stage(ctypes.Structure):
    _fields_ = [("blocks", ctypes.POINTER(gsn_block)), ("n_blocks", c_int), ("W", c_vp), ("bias", c_vp), ("n_out", c_i64),
                ("bn_mean", c_vp), ("bn_scale", c_vp), ("bn_shift", c_vp), ("act", c_int)]


class gsn_self_block(ctypes.Structure):
    _fields_ = [("data", c_vp), ("width", c_i64), ("row_stride", c_i64)]


class gsn_pack16(ctypes.Structure):
    _fields_ = [("node_rows", c_vp), ("edge_rows", c_vp)]


class gsn_count_side(ctypes.Structure):
    _fields_ = [("csr_row", c_int), ("seg_ptr", c_vp), ("perm", c_vp), ("sorted_target", c_vp), ("sorted_other", c_vp),
                ("n_nodes", c_i64), ("n_edges", c_i64), ("node_codes", c_vp), ("node_code_cols", c_int), ("node_n_classes", c_int * 4),
                ("node_clamp", c_int), ("node_pack", c_vp), ("edge_codes", c_vp), ("edge_code_cols", c_int), ("edge_n_classes", c_int * 4),
                ("edge_clamp", c_int), ("edge_col0", c_int), ("code_status", c_vp)]


class gsn_count_call(ctypes.Structure):
    _fields_ = [("plan_host", c_vp), ("plan_dev", c_vp), ("plan_words", c_i64), ("n_graphs", c_i64), ("node_ptr", c_vp), ("edge_ptr", c_vp),
                ("edge_index", c_vp), ("edge_row_stride", c_i64), ("ids_are_global", c_int), ("max_nodes", c_i64), ("max_edges", c_i64),
                ("out", c_vp), ("status", c_vp), ("n_classes", c_vp), ("clamp", c_int), ("pack", c_vp), ("pack_stride", c_i64),
                ("pack_col0", c_i64), ("side", ctypes.POINTER(gsn_count_side))]


class gsn_layer_pack16_call(ctypes.Structure):
    _fields_ = [("n_nodes", c_i64), ("n_edges", c_i64), ("seg_ptr", c_vp), ("edge", ctypes.POINTER(gsn_chain_stage)), ("x", c_vp), ("d_x", c_i64),
                ("node0", ctypes.POINTER(gsn_chain_stage)), ("node1", ctypes.POINTER(gsn_chain_stage)), ("prepared", c_vp),
                ("pack", ctypes.POINTER(gsn_pack16)), ("edge_rows", c_i64), ("out", c_vp)]


# name -> (restype, argtypes); kept in one place so tests can check it against include/gsn_abi.h
class gsn_code_slot(ctypes.Structure):
    _fields_ = [("codes", ctypes.c_void_p), ("idx", ctypes.c_void_p), ("stride", ctypes.c_int32), ("col", ctypes.c_int32),
                ("w_off", ctypes.c_int32), ("n_classes", ctypes.c_int32), ("clamp", ctypes.c_int32), ("reserved", ctypes.c_int32)]


SIGNATURES = {
    "gsn_last_error": (ctypes.c_char_p, []),
    "gsn_version": (c_int, []),
    "gsn_device_count": (c_int, []),
    "gsn_stream_capture_id": (c_i64, [c_vp]),
    "gsn_fingerprint_hip": (c_int, [c_int, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "gsn_pattern_orbits": (c_int, [c_i64, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gsn_graph_vertex_orbits": (c_int, [c_i64, c_i64, c_vp, c_vp, c_vp]),
    "gsn_count_plan_build": (c_int, [c_int, c_int, c_int, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "gsn_count_hip": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_i64,
                              c_vp, c_vp, c_vp]),
    "gsn_count_encode_hip": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_i64,
                                     c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "gsn_count_encode_pack16_hip": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_i64,
                                            c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "gsn_count_encode_pack16_side_hip": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_i64, c_i64, c_vp, c_vp, c_vp, c_int,
                                                 c_vp, c_i64, c_i64, ctypes.POINTER(gsn_count_side), c_vp]),
    "gsn_count_layer_step_hip": (c_int, [ctypes.POINTER(gsn_count_call), ctypes.POINTER(gsn_layer_pack16_call), c_vp, c_vp]),
    "gsn_pack16_rows_hip": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "gsn_one_hot_pack16_hip": (c_int, [c_i64, c_int, c_vp, c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "gsn_csr_scratch_elems": (c_i64, [c_i64]),
    "gsn_csr_build_hip": (c_int, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gsn_linear_f16x3_kpad": (c_i64, [c_i64]),
    "gsn_linear_f16x3_mpad": (c_i64, [c_i64]),
    "gsn_linear_f16x3_scratch_bytes": (c_i64, [c_i64, c_i64]),
    "gsn_linear_f16x3_prepare_hip": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "gsn_linear_f16x3_prepare_strided_hip": (c_int, [c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "gsn_linear_fwd_strided_hip": (c_int, [c_i64, c_int, ctypes.POINTER(gsn_block), c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_int,
                                           c_vp, c_vp, c_vp]),
    "gsn_linear_splitk_plan": (c_int, [c_i64, c_i64, c_i64]),
    "gsn_linear_fwd_splitk_hip": (c_int, [c_i64, c_int, ctypes.POINTER(gsn_block), c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "gsn_linear_f16x3_fwd_hip": (c_int, [c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "gsn_linear_f16x3_fwd_stats_hip": (c_int, [c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "gsn_linear_f16x3_split_rows_hip": (c_int, [c_i64, c_int, c_vp, c_vp, c_vp]),
    "gsn_linear_f16x3_fwd_presplit_hip": (c_int, [c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "gsn_wgrad_f16x3_hip": (c_int, [c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "gsn_linear_f16x3_fwd_stats_presplit_hip": (c_int, [c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "gsn_bn_act_planes_hip": (c_int, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "gsn_edge_split_sum_hip": (c_int, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp]),
    "gsn_csr_build_graphs_hip": (c_int, [c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gsn_segsum_prepare_hip": (c_int, [c_i64, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "gsn_segment_sum_rows_hip": (c_int, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "gsn_propagate_fwd_hip": (c_int, [c_int, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_vp,
                                      c_i64, c_vp, c_vp]),
    "gsn_propagate_bwd_hip": (c_int, [c_int, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_int,
                                      c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gsn_propagate_self_fwd_hip": (c_int, [c_int, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_vp,
                                           c_i64, c_i64, c_i64, c_int, ctypes.POINTER(gsn_self_block), c_vp, c_vp, c_vp]),
    "gsn_add_gathered_hip": (c_int, [c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "gsn_propagate_self_bwd_hip": (c_int, [c_int, c_i64, c_i64, c_vp, c_int, ctypes.POINTER(gsn_self_block), ctypes.POINTER(c_vp), c_vp, c_vp,
                                           c_vp, c_vp]),
    "gsn_propagate_pad_bwd_hip": (c_int, [c_int, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_int,
                                          c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gsn_propagate_bwd_fold_self_hip": (c_int, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp,
                                                c_vp, c_vp, c_vp]),
    "gsn_linear_fwd_hip": (c_int, [c_i64, c_int, ctypes.POINTER(gsn_block), c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_int,
                                   c_vp, c_vp, c_vp, c_vp]),
    "gsn_one_hot_hip": (c_int, [c_i64, c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
    "gsn_column_range_hip": (c_int, [c_i64, c_int, c_vp, c_vp, c_vp, c_vp]),
    "gsn_column_ranks_hip": (c_int, [c_i64, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "gsn_embed_fwd_hip": (c_int, [c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gsn_embed_bwd_hip": (c_int, [c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gsn_embed_bwd_flat_supported": (c_int, [c_i64, c_int, c_int, c_vp]),
    "gsn_embed_bwd_flat_hip": (c_int, [c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gsn_code_stage_supported": (c_int, [c_int, c_i64, c_i64]),
    "gsn_code_stage_fwd_hip": (c_int, [c_i64, c_int, ctypes.POINTER(gsn_code_slot), c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp,
                                       c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gsn_bn_act_hip": (c_int, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "gsn_bn_act_bwd_hip": (c_int, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "gsn_fold_weights_fwd_hip": (c_int, [c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "gsn_fold_weights_bwd_hip": (c_int, [c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gsn_bn_act_bwd_from_h_hip": (c_int, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "gsn_bn_act_bwd_planes_hip": (c_int, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "gsn_wgrad_hip": (c_int, [c_i64, c_i64, c_vp, c_int, ctypes.POINTER(gsn_block), c_vp, c_vp]),
    "gsn_gather_cat_hip": (c_int, [c_i64, c_int, ctypes.POINTER(gsn_block), c_vp, c_vp]),
    "gsn_bn_finalize_hip": (c_int, [c_i64, c_i64, ctypes.c_double, ctypes.c_double, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "gsn_bn_finalize_act_hip": (c_int, [c_i64, c_i64, ctypes.c_double, ctypes.c_double, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_vp, c_int, c_vp, c_vp]),
    "gsn_bn_finalize_count_hip": (c_int, [c_i64, c_i64, ctypes.c_double, ctypes.c_double, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                          c_vp]),
    "gsn_column_stats_hip": (c_int, [c_i64, c_i64, c_vp, c_vp, c_vp]),
    "gsn_mlp_chain_supported": (c_int, [c_int, ctypes.POINTER(gsn_chain_stage)]),
    "gsn_layer_fused_supported": (c_int, [ctypes.POINTER(gsn_chain_stage), c_i64, ctypes.POINTER(gsn_chain_stage), ctypes.POINTER(gsn_chain_stage)]),
    "gsn_layer_fused_prepared_bytes": (c_i64, [ctypes.POINTER(gsn_chain_stage), c_i64, ctypes.POINTER(gsn_chain_stage), ctypes.POINTER(gsn_chain_stage)]),
    "gsn_layer_fused_prepare_hip": (c_int, [ctypes.POINTER(gsn_chain_stage), c_i64, ctypes.POINTER(gsn_chain_stage), ctypes.POINTER(gsn_chain_stage),
                                            c_vp, c_vp]),
    "gsn_layer_fused_fwd_hip": (c_int, [c_i64, c_i64, c_vp, ctypes.POINTER(gsn_chain_stage), c_vp, c_i64, ctypes.POINTER(gsn_chain_stage),
                                        ctypes.POINTER(gsn_chain_stage), c_vp, c_vp, c_vp]),
    "gsn_layer_fused_pack16_supported": (c_int, [ctypes.POINTER(gsn_chain_stage), c_i64, ctypes.POINTER(gsn_chain_stage), ctypes.POINTER(gsn_chain_stage)]),
    "gsn_layer_fused_pack16_prepared_bytes": (c_i64, [ctypes.POINTER(gsn_chain_stage), c_i64, ctypes.POINTER(gsn_chain_stage), ctypes.POINTER(gsn_chain_stage)]),
    "gsn_layer_fused_pack16_prepare_hip": (c_int, [ctypes.POINTER(gsn_chain_stage), c_i64, ctypes.POINTER(gsn_chain_stage), ctypes.POINTER(gsn_chain_stage), c_vp, c_vp]),
    "gsn_layer_fused_fwd_pack16_hip": (c_int, [c_i64, c_i64, c_vp, ctypes.POINTER(gsn_chain_stage), c_vp, c_i64, ctypes.POINTER(gsn_chain_stage), ctypes.POINTER(gsn_chain_stage), c_vp, ctypes.POINTER(gsn_pack16), c_i64, c_vp, c_vp]),
    "gsn_layer_fused_workspace_bytes": (c_i64, [c_i64, ctypes.POINTER(gsn_chain_stage), c_i64, ctypes.POINTER(gsn_chain_stage), ctypes.POINTER(gsn_chain_stage)]),
    "gsn_layer_fused_fwd_ws_hip": (c_int, [c_i64, c_i64, c_vp, ctypes.POINTER(gsn_chain_stage), c_vp, c_i64, ctypes.POINTER(gsn_chain_stage),
                                           ctypes.POINTER(gsn_chain_stage), c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "gsn_layer_fused_graphs_supported": (c_int, [ctypes.POINTER(gsn_chain_stage), c_i64, ctypes.POINTER(gsn_chain_stage), ctypes.POINTER(gsn_chain_stage)]),
    "gsn_layer_fused_fwd_graphs_hip": (c_int, [c_i64, c_i64, c_vp, ctypes.POINTER(gsn_chain_stage), c_vp, c_i64, ctypes.POINTER(gsn_chain_stage),
                                               ctypes.POINTER(gsn_chain_stage), c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "gsn_mlp_chain_fwd_hip": (c_int, [c_i64, c_int, ctypes.POINTER(gsn_chain_stage), c_vp, c_vp, c_vp, c_vp, c_vp]),
}


def build(verbose: bool = False) -> str:
    """Compile libgsn_hip.so for gfx950 (idempotent: make only rebuilds what changed)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libgsn_hip.so failed:\n" + res.stdout)
    if verbose:
        print(res.stdout)
    return LIB_PATH


def lib():
    """The loaded library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "gsn_amd: %s not found. Build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C gsn_amd/csrc). There is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP/HSA runtime; it must be the one already loaded when libgsn_hip.so is
        # opened so that both share one runtime (two HSA runtimes in a process cannot both own the device).
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the build is stale: loud by design
            fn.restype = res
            fn.argtypes = args
        if handle.gsn_version() != 1:
            raise ImportError("gsn_amd: libgsn_hip.so ABI version %d != 1" % handle.gsn_version())
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().gsn_last_error()
        raise GsnError("%s failed (%d): %s" % (what or "libgsn_hip", rc, msg.decode() if msg else ""))


_GPU_SEEN = False


def require_gpu() -> None:
    """Device entry points need a real gfx950; refuse loudly otherwise.  (torch.cuda.is_available() costs ~40 us a call: a GPU that
    has been seen once stays.)"""
    global _GPU_SEEN
    if _GPU_SEEN:
        return
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("gsn_amd: no GPU visible to PyTorch-ROCm; the counting / message-passing kernels are "
                           "HIP-only (gfx950) and there is no CPU fallback.")
    _GPU_SEEN = True


def ptr(t) -> int:
    """data pointer of a torch tensor / numpy array / None"""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def current_stream() -> int:
    """Raw handle of PyTorch's current stream on the current device (the fast C getter: this runs before every launch)."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class _NullGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL_GUARD = _NullGuard()


def device_guard(device):
    """``torch.cuda.device(device)`` only when ``device`` is not already current (the context manager costs ~10 us)."""
    import torch
    idx = device.index
    if idx is None or idx == torch._C._cuda_getDevice():
        return _NULL_GUARD
    return torch.cuda.device(device)

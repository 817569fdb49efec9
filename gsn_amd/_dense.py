"""Forward dispatch of the dense stages ``act(bn(cat(blocks) W^T + b))`` (models_misc.py:41-59) and of the one-launch layers: which kernel
takes a stage list (chain kernels, fp16x3 / bf16x6 linear kernels, the one-launch layer kernels), train-mode BatchNorm statistics."""
from __future__ import annotations

import ctypes
import os

import torch

from . import _abi, flags
from ._caches import _note_cache
from ._runtime import _ACT_CODE, _MAX_BLOCKS, _f32c, _timed, _zeros

# ------------------------------------------------------------------------------------------------------------------
# fused Linear (+BN) (+activation) stage
# ------------------------------------------------------------------------------------------------------------------


def _f16x3_weights(weight, w32):
    """fp16 planes + inverse column scales of a weight matrix for gsn_linear_f16x3_fwd_hip, made once per weight VERSION and kept
    on the tensor object (parameters, folded weights and the cached derived matrices all live across calls)."""
    key = (weight._version, w32.data_ptr(), tuple(w32.shape))
    hit = getattr(weight, "_gsn_f16x3", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    n_out, k = w32.shape
    L = _abi.lib()
    kpad = int(L.gsn_linear_f16x3_kpad(k))
    planes = torch.empty(2 * n_out * kpad, dtype=torch.float16, device=w32.device)
    col_inv = torch.empty(n_out, dtype=torch.float32, device=w32.device)
    with _abi.device_guard(w32.device):
        if w32.is_contiguous():
            _abi.check(L.gsn_linear_f16x3_prepare_hip(w32.data_ptr(), n_out, k, planes.data_ptr(), col_inv.data_ptr(), _abi.current_stream()),
                       "gsn_linear_f16x3_prepare_hip")
        else:       # (a transposed view: read through its strides, no copy)
            _abi.check(L.gsn_linear_f16x3_prepare_strided_hip(w32.data_ptr(), n_out, k, w32.stride(0), w32.stride(1), planes.data_ptr(), col_inv.data_ptr(),
                                                              _abi.current_stream()), "gsn_linear_f16x3_prepare_strided_hip")
    try:
        weight._gsn_f16x3 = (key, planes, col_inv)
        _note_cache(weight, "_gsn_f16x3")
    except (AttributeError, RuntimeError):
        pass
    return planes, col_inv


def _f16x3_takes(m_rows, n_out, widths, aligned=True):
    """Would _linear_hip run this product of direct rows (block widths ``widths``) on the fp16x3 kernel?  (the rule inside it, for callers that
    decide what to keep or prepare for that kernel)"""
    return (flags.LINEAR_F16X3 and m_rows > 0 and n_out > flags.LINEAR_F16X3_MIN_N
            and ((m_rows + 127) // 128) * ((n_out + 127) // 128) > flags.LINEAR_F16X3_MIN_TILES and aligned and all(w % 4 == 0 for w in widths))


def _linear_hip(blocks, weight, bias, bn_mean, bn_scale, bn_shift, act, m_rows, out=True, stats=None, split_k=False, scratch_out=None, presplit=None):
    """blocks: list of (data [R,w] fp32 cuda, idx int64 [M] or None).  ``split_k``: the caller accepts a sum whose order varies from run to run in the
    last bits (float atomics over K ranges, gsn_linear_fwd_splitk_hip) -- the input-gradient products of a backward pass, whose weight gradients
    are accumulated that way already; forward products stay on one workgroup per tile: an eval forward is reproducible bit for bit.
    ``scratch_out``: a list that receives the row scratch (inverse row scales + the rows' fp16 planes) when the product ran on the fp16x3 kernel --
    what gsn_wgrad_f16x3_hip multiplies (the caller keeps it alive).  ``presplit``: the row scratch of the rows, made earlier (gsn_bn_act_bwd_planes_hip):
    the product on the fp16x3 kernel without its pre-pass; ``blocks`` then only name the widths (any 16-byte aligned fp32 tensor of that shape)."""
    if len(blocks) > _MAX_BLOCKS:
        raise NotImplementedError("more than %d input blocks" % _MAX_BLOCKS)
    dev = weight.device
    arr = (_abi.gsn_block * len(blocks))()
    keep = []
    for i, (d, idx) in enumerate(blocks):
        d = _f32c(d)
        keep.append(d)
        arr[i].data = d.data_ptr()
        arr[i].idx = None
        arr[i].idx32 = None
        if idx is not None:
            idx = idx.contiguous()
            keep.append(idx)
            if idx.dtype == torch.int32:
                arr[i].idx32 = idx.data_ptr()
            else:
                arr[i].idx = idx.data_ptr()
        arr[i].width = d.shape[1]
    n_out = weight.shape[0]
    y = torch.empty((m_rows, n_out), dtype=torch.float32, device=dev) if out else None
    if m_rows == 0:
        return y                      # (no rows: nothing to launch; `stats` keeps its zeros)
    # a transposed VIEW of a row-major fp32 matrix (the input-gradient product gX = gH W reads the stage's weight as its transpose) is taken
    # through its strides by both dense kernels: no transposed copy per stage and step
    w_view = (weight.dim() == 2 and weight.dtype is torch.float32 and not weight.is_contiguous() and weight.stride(0) == 1
              and weight.stride(1) >= weight.shape[0] and flags.STRIDED_WEIGHTS)
    w = (weight.detach() if weight.requires_grad else weight) if w_view else _f32c(weight)
    vecs = [None if v is None else _f32c(v) for v in (bias, bn_mean, bn_scale, bn_shift)]
    # direct rows (node-level stages): the fp16x3 kernel with the weights split once per weight version
    # (from two column tiles on: the pre-pass over the rows that finds their scales is then amortised -- at n_out <= 128 the
    #  bf16x6 kernel, which reads the rows once, is faster: 99 vs 90 TF/s at K = 260)
    # (a train-mode stage that keeps its pre-BN rows: the same kernel with the column statistics taken in its epilogue)
    takes16 = (flags.LINEAR_F16X3 and out and (stats is None or (flags.LINEAR_F16X3_STATS and bn_mean is None and act == 0 and n_out % 4 == 0)) and m_rows > 0 and n_out > flags.LINEAR_F16X3_MIN_N
               and ((m_rows + 127) // 128) * ((n_out + 127) // 128) > flags.LINEAR_F16X3_MIN_TILES
               and all(idx is None for _, idx in blocks) and all(d.shape[1] % 4 == 0 and d.data_ptr() % 16 == 0 for d in keep))
    if presplit is not None and not takes16:
        raise RuntimeError("presplit rows given to a product that does not run on the fp16x3 kernel (_f16x3_takes decides)")
    if takes16:
        planes, col_inv = _f16x3_weights(weight, w)
        scratch = presplit if presplit is not None else torch.empty(int(_abi.lib().gsn_linear_f16x3_scratch_bytes(m_rows, w.shape[1])), dtype=torch.uint8, device=dev)
        with _abi.device_guard(dev), _timed("linear_fwd", 2.0 * m_rows * w.shape[1] * n_out):
            if presplit is not None and stats is not None:
                rc = _abi.lib().gsn_linear_f16x3_fwd_stats_presplit_hip(m_rows, len(blocks), arr, planes.data_ptr(), col_inv.data_ptr(), _abi.ptr(vecs[0]), n_out,
                                                                        scratch.data_ptr(), y.data_ptr(), stats.data_ptr(), _abi.current_stream())
            elif presplit is not None:
                rc = _abi.lib().gsn_linear_f16x3_fwd_presplit_hip(m_rows, len(blocks), arr, planes.data_ptr(), col_inv.data_ptr(), _abi.ptr(vecs[0]), n_out,
                                                                  _abi.ptr(vecs[1]), _abi.ptr(vecs[2]), _abi.ptr(vecs[3]), act, scratch.data_ptr(),
                                                                  y.data_ptr(), _abi.current_stream())
            elif stats is not None:
                rc = _abi.lib().gsn_linear_f16x3_fwd_stats_hip(m_rows, len(blocks), arr, planes.data_ptr(), col_inv.data_ptr(), _abi.ptr(vecs[0]), n_out,
                                                               scratch.data_ptr(), y.data_ptr(), stats.data_ptr(), _abi.current_stream())
            else:
                rc = _abi.lib().gsn_linear_f16x3_fwd_hip(m_rows, len(blocks), arr, planes.data_ptr(), col_inv.data_ptr(), _abi.ptr(vecs[0]), n_out,
                                                         _abi.ptr(vecs[1]), _abi.ptr(vecs[2]), _abi.ptr(vecs[3]), act, scratch.data_ptr(),
                                                         y.data_ptr(), _abi.current_stream())
        _abi.check(rc, "gsn_linear_f16x3_fwd_presplit_hip" if presplit is not None else ("gsn_linear_f16x3_fwd_stats_hip" if stats is not None else "gsn_linear_f16x3_fwd_hip"))
        if scratch_out is not None:
            scratch_out.append(scratch)
        return y
    # few rows, identity epilogue (the input-gradient products of a dense backward at the reference's batch sizes): the K slices of an output tile
    # shared by up to four workgroups that add into a zero-filled output (from the zero arena: no fill launch per product)
    if split_k and out and stats is None and act == 0 and bn_mean is None and bn_scale is None and flags.LINEAR_SPLITK:
        splits = int(_abi.lib().gsn_linear_splitk_plan(m_rows, w.shape[1], n_out))
        if splits > 1:
            from ._runtime import _zeros
            y = _zeros(m_rows * n_out, torch.float32, dev).view(m_rows, n_out)
            with _abi.device_guard(dev), _timed("linear_fwd", 2.0 * m_rows * w.shape[1] * n_out):
                rc = _abi.lib().gsn_linear_fwd_splitk_hip(m_rows, len(blocks), arr, w.data_ptr(), w.stride(0) if w_view else 0, w.stride(1) if w_view else 0,
                                                          _abi.ptr(vecs[0]), n_out, y.data_ptr(), _abi.current_stream())
            _abi.check(rc, "gsn_linear_fwd_splitk_hip")
            return y
    if w_view:
        with _abi.device_guard(dev), _timed("linear_fwd", 2.0 * m_rows * w.shape[1] * n_out):
            rc = _abi.lib().gsn_linear_fwd_strided_hip(m_rows, len(blocks), arr, w.data_ptr(), w.stride(0), w.stride(1), _abi.ptr(vecs[0]), n_out,
                                                       _abi.ptr(vecs[1]), _abi.ptr(vecs[2]), _abi.ptr(vecs[3]), act, _abi.ptr(y), _abi.ptr(stats),
                                                       _abi.current_stream())
        if rc != -2:                  # (GSN_E_UNSUPPORTED: the bf16x6 kernel is switched off -> a contiguous copy below)
            _abi.check(rc, "gsn_linear_fwd_strided_hip")
            return y
        w = w.contiguous()
    with _abi.device_guard(dev), _timed("linear_fwd", 2.0 * m_rows * w.shape[1] * n_out):
        rc = _abi.lib().gsn_linear_fwd_hip(m_rows, len(blocks), arr, w.data_ptr(), _abi.ptr(vecs[0]), n_out, _abi.ptr(vecs[1]),
                                           _abi.ptr(vecs[2]), _abi.ptr(vecs[3]), act, None, _abi.ptr(y), _abi.ptr(stats),
                                           _abi.current_stream())
    _abi.check(rc, "gsn_linear_fwd_hip")
    return y


class _Stage:
    """One Linear (+BatchNorm1d) (+activation) stage: ``act(bn([blocks | previous output] W^T + b))``."""
    __slots__ = ("blocks", "weight", "bias", "bn", "act", "bn_params", "bn_invstd")

    def __init__(self, weight, bias, bn=None, act="identity", blocks=()):
        self.blocks, self.weight, self.bias, self.bn, self.act = list(blocks), weight, bias, bn, act
        self.bn_params = None  # (mean, scale, shift) once resolved
        self.bn_invstd = None  # batch invstd of a train-mode stage (for the adjoint)


def _launch_stages(stages, m_rows, stats=None, csr=None):
    """Run resolved stages: fused gsn_mlp_chain_fwd_hip where it fits (<= 2 stages per launch), else stage by stage.
    With ``csr`` the rows are visited in target-sorted order and the LAST stage's rows are summed per target (fused
    scatter-add) -> [n_nodes, n_out]; returns None if that cannot be fused (caller falls back to propagate)."""
    L = _abi.lib()
    dev = stages[0].weight.device
    if m_rows == 0:
        # no rows (an edge-less batch in front of an edge stage): nothing to launch -- empty output, zero sums per target
        n_last = stages[-1].weight.shape[0]
        if csr is not None:
            return torch.zeros((csr.seg_ptr.numel() - 1, n_last), dtype=torch.float32, device=dev)
        return torch.empty((0, n_last), dtype=torch.float32, device=dev) if stats is None else None
    y = None
    i = 0
    while i < len(stages):
        group = None
        for n in (2, 1):
            cand = stages[i:i + n]
            if len(cand) < n:
                continue
            carry = [(y, None)] if y is not None else []
            arr = (_abi.gsn_chain_stage * n)()
            keep = []
            for j, st in enumerate(cand):
                blks = (carry if j == 0 else []) + st.blocks
                if j == 0 and y is not None:
                    blks = st.blocks + carry          # concatenation order: own HBM blocks, then the previous output
                barr = (_abi.gsn_block * max(len(blks), 1))()
                for b, (d, idx) in enumerate(blks):
                    d = _f32c(d); keep.append(d)
                    barr[b].data = d.data_ptr(); barr[b].width = d.shape[1]
                    barr[b].idx = None; barr[b].idx32 = None
                    if idx is not None:
                        idx = idx.contiguous(); keep.append(idx)
                        if idx.dtype == torch.int32:
                            barr[b].idx32 = idx.data_ptr()
                        else:
                            barr[b].idx = idx.data_ptr()
                keep.append(barr)
                w = _f32c(st.weight); keep.append(w)
                arr[j].blocks = barr; arr[j].n_blocks = len(blks)
                arr[j].W = w.data_ptr(); arr[j].n_out = w.shape[0]
                vecs = [None if v is None else _f32c(v) for v in ((st.bias,) + (st.bn_params or (None, None, None)))]
                keep.extend(vecs)
                arr[j].bias, arr[j].bn_mean, arr[j].bn_scale, arr[j].bn_shift = [_abi.ptr(v) for v in vecs]
                arr[j].act = _ACT_CODE[st.act]
            if L.gsn_mlp_chain_supported(n, arr):
                group = (n, arr, keep, cand)
                break
        last_group = group is not None and i + group[0] == len(stages)
        if csr is not None and not (last_group and i == 0):
            return None   # the fused scatter-add needs the whole stage list in one launch
        if group is not None:
            n, arr, keep, cand = group
            want_stats = stats is not None and last_group
            n_out = cand[-1].weight.shape[0]
            seg = csr is not None and not want_stats
            if seg:
                n_seg = csr.seg_ptr.numel() - 1
                out = torch.empty((n_seg, n_out), dtype=torch.float32, device=dev)
                with _abi.device_guard(dev), _timed("segsum_prepare"):
                    _abi.check(L.gsn_segsum_prepare_hip(n_seg, m_rows, csr.seg_ptr.data_ptr(), csr.tgt.data_ptr(), n_out,
                                                        out.data_ptr(), _abi.current_stream()), "gsn_segsum_prepare_hip")
            else:
                out = None if want_stats else torch.empty((m_rows, n_out), dtype=torch.float32, device=dev)
            flops = 0.0
            kprev = 0
            for j, st in enumerate(cand):
                flops += 2.0 * m_rows * st.weight.shape[1] * st.weight.shape[0]
            with _abi.device_guard(dev), _timed("mlp_chain%d" % n, flops):
                rc = L.gsn_mlp_chain_fwd_hip(m_rows, n, arr, None,
                                             csr.tgt.data_ptr() if seg else None, _abi.ptr(out),
                                             _abi.ptr(stats) if want_stats else None, _abi.current_stream())
            _abi.check(rc, "gsn_mlp_chain_fwd_hip")
            y = out
            i += n
        else:
            st = stages[i]
            blks = st.blocks + ([(y, None)] if y is not None else [])
            last = i == len(stages) - 1
            bp = st.bn_params or (None, None, None)
            if last and stats is not None:
                _linear_hip(blks, st.weight, st.bias, None, None, None, 0, m_rows, out=False, stats=stats)
                y = None
            else:
                y = _linear_hip(blks, st.weight, st.bias, bp[0], bp[1], bp[2], _ACT_CODE[st.act], m_rows)
            i += 1
    return y


def _chain_fits(stages):
    """True if the whole stage list runs as ONE gsn_mlp_chain_fwd_hip launch (needed for the fused scatter-add)."""
    n = len(stages)
    if n < 1 or n > 2:
        return False
    arr = (_abi.gsn_chain_stage * n)()
    keep = []
    for j, st in enumerate(stages):
        barr = (_abi.gsn_block * max(len(st.blocks), 1))()
        for b, (d, idx) in enumerate(st.blocks):
            barr[b].data = 1; barr[b].idx = None; barr[b].idx32 = None; barr[b].width = d.shape[1]
        keep.append(barr)
        arr[j].blocks = barr; arr[j].n_blocks = len(st.blocks)
        arr[j].W = 1; arr[j].n_out = st.weight.shape[0]; arr[j].act = _ACT_CODE[st.act]
    return bool(_abi.lib().gsn_mlp_chain_supported(n, arr))


def _stage_struct(st, blocks, keep):
    """gsn_chain_stage of a resolved _Stage (BN parameters already in st.bn_params)."""
    g = _abi.gsn_chain_stage()
    barr = (_abi.gsn_block * max(len(blocks), 1))()
    for b, (d, idx) in enumerate(blocks):
        d = _f32c(d); keep.append(d)
        barr[b].data = d.data_ptr(); barr[b].width = d.shape[1]
        barr[b].idx = None; barr[b].idx32 = None
        if idx is not None:
            idx = idx.contiguous(); keep.append(idx)
            if idx.dtype == torch.int32:
                barr[b].idx32 = idx.data_ptr()
            else:
                barr[b].idx = idx.data_ptr()
    keep.append(barr)
    w = _f32c(st.weight); keep.append(w)
    g.blocks = barr; g.n_blocks = len(blocks)
    g.W = w.data_ptr(); g.n_out = w.shape[0]
    vecs = [None if v is None else _f32c(v) for v in ((st.bias,) + (st.bn_params or (None, None, None)))]
    keep.extend(vecs)
    g.bias, g.bn_mean, g.bn_scale, g.bn_shift = [_abi.ptr(v) for v in vecs]
    g.act = _ACT_CODE[st.act]
    return g


def _prep_key(st):
    bn = st.bn
    bk = None
    if bn is not None:
        bk = (bn.running_mean._version, bn.running_var._version, bn.running_mean.data_ptr(),
              (bn.weight._version, bn.bias._version) if bn.affine else None)
    return (st.weight.data_ptr(), st.weight._version, tuple(st.weight.shape), bk)


def _layer_fused(x, csr, edge_stages, node_stages, training, owner=None, gen=0, pack16=None, pack_only=False, record=None):
    """edge stage + per-target sum + two node stages in ONE launch (gsn_layer_fused_fwd_hip); None if the layer does not
    fit (shape, activation, or a BatchNorm1d that needs batch statistics)."""
    if not flags.FUSED_LAYER or len(edge_stages) != 1 or len(node_stages) != 2:
        return None
    stages = edge_stages + node_stages
    for st in stages:
        if st.act not in ("identity", "relu"):
            return None
        # (a BatchNorm1d that is itself in train mode takes batch statistics whatever the layer's flag says: not this kernel's arithmetic)
        if st.bn is not None and (training or st.bn.training or st.bn.running_mean is None):
            return None
    if x.data_ptr() % 16:
        return None
    if len(node_stages[0].blocks) != 1 or node_stages[1].blocks:     # ([x | S | deg]: S and deg are produced inside the kernel)
        return None
    for st in stages:
        _bn_resolve(st, None, 0, False)
    keep = []
    ge = _stage_struct(edge_stages[0], edge_stages[0].blocks, keep)
    g0 = _stage_struct(node_stages[0], [], keep)
    g1 = _stage_struct(node_stages[1], [], keep)
    L = _abi.lib()
    d_x = x.shape[1]
    if node_stages[0].weight.shape[1] != d_x + edge_stages[0].weight.shape[0] + 4:
        return None
    n = x.shape[0]
    E = csr.tgt.numel()
    flops = 2.0 * E * edge_stages[0].weight.shape[1] * edge_stages[0].weight.shape[0]
    flops += 2.0 * n * (node_stages[0].weight.shape[1] * node_stages[0].weight.shape[0] + node_stages[1].weight.shape[1] * node_stages[1].weight.shape[0])
    # tagged exact inputs (gsn_amd.packs): the same layer on their fp16 packs -- own prepared weights (another k-slot order), kept beside
    # the fp32 kernel's under their own key
    if pack16 is not None and flags.PACK16_LAYER and L.gsn_layer_fused_pack16_supported(ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1)):
        key = (tuple(_prep_key(st) for st in stages), d_x, gen, "pack16")
        hit = getattr(owner, "_fused_prep16", None) if owner is not None else None
        if hit is not None and hit[0] == key:
            prep = hit[1]
        else:
            nbytes = int(L.gsn_layer_fused_pack16_prepared_bytes(ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1)))
            prep = torch.empty(nbytes // 4, dtype=torch.int32, device=x.device)
            with _abi.device_guard(x.device):
                _abi.check(L.gsn_layer_fused_pack16_prepare_hip(ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1), prep.data_ptr(),
                                                                _abi.current_stream()), "gsn_layer_fused_pack16_prepare_hip")
            if owner is not None:
                owner._fused_prep16 = (key, prep)
                _note_cache(owner, "_fused_prep16")
        n_out = node_stages[1].weight.shape[0]
        pk = _abi.gsn_pack16()
        fl_e = 2.0 * edge_stages[0].weight.shape[1] * edge_stages[0].weight.shape[0]
        fl_n = 2.0 * (node_stages[0].weight.shape[1] * node_stages[0].weight.shape[0] + node_stages[1].weight.shape[1] * n_out)

        def launch16(x_ptr, n_rows, dev, csr_now, idx_now, pack_now):
            """The recorded call with this forward's pointers (x placeholder / rows, block indices, packs): one foreign call, no Python-side
            stage building.  None when the library declines (packs beyond 32-bit offsets)."""
            nb = ge.n_blocks
            for i in range(nb):
                ge.blocks[i].data = x_ptr if i < 2 else pack_now[1].data_ptr()
                ge.blocks[i].idx32 = idx_now[i].data_ptr()
            y = torch.empty((n_rows, n_out), dtype=torch.float32, device=dev)
            pk.node_rows = pack_now[0].data_ptr()
            pk.edge_rows = None if pack_now[1] is None else pack_now[1].data_ptr()
            e_now = csr_now.tgt.numel()
            with _abi.device_guard(dev), _timed("layer_fused", fl_e * e_now + fl_n * n_rows):
                rc16 = L.gsn_layer_fused_fwd_pack16_hip(n_rows, e_now, csr_now.seg_ptr.data_ptr(), ctypes.byref(ge), x_ptr, d_x, ctypes.byref(g0), ctypes.byref(g1),
                                                        prep.data_ptr(), ctypes.byref(pk), 0 if pack_now[1] is None else pack_now[1].shape[0], y.data_ptr(),
                                                        _abi.current_stream())
            if rc16 == -2:
                return None
            _abi.check(rc16, "gsn_layer_fused_fwd_pack16_hip")
            return y
        launch16._keep = keep
        out = launch16(x.data_ptr(), n, x.device, csr, [b[1] for b in edge_stages[0].blocks], pack16)
        if out is not None:               # (None = GSN_E_UNSUPPORTED: packs beyond 32-bit offsets -> the fp32 kernel below)
            if record is not None and owner is not None and all(b[1] is not None and b[1].dtype == torch.int32 for b in edge_stages[0].blocks):
                owner._fplan = (record, "pack16", launch16)
                _note_cache(owner, "_fplan")
            return out
    if pack_only:                         # (the caller holds no fp32 rows: it makes them and comes back)
        return None
    # the weights as the kernel's register fragments: once per weight version (kept on the layer module)
    if not L.gsn_layer_fused_supported(ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1)):
        return None
    out = torch.empty((n, node_stages[1].weight.shape[0]), dtype=torch.float32, device=x.device)
    # (the kernel variant the buffer is for -- this file's kernel alone, with the register-resident fragments appended, the d = 128
    #  layout -- follows from the block properties of THIS call: its size is part of the key)
    nbytes = int(L.gsn_layer_fused_prepared_bytes(ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1)))
    key = (tuple(_prep_key(st) for st in stages), d_x, gen, nbytes)
    hit = getattr(owner, "_fused_prep", None) if owner is not None else None
    if hit is not None and hit[0] == key:
        prep = hit[1]
    else:
        prep = torch.empty(nbytes // 4, dtype=torch.int32, device=x.device)
        with _abi.device_guard(x.device):
            _abi.check(L.gsn_layer_fused_prepare_hip(ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1), prep.data_ptr(),
                                                     _abi.current_stream()), "gsn_layer_fused_prepare_hip")
        if owner is not None:
            owner._fused_prep = (key, prep)
            _note_cache(owner, "_fused_prep")
    # a collated batch with known graph boundaries, every graph <= 128 vertices: the d = 128 layer on graph-aligned tiles (csrc/layer_g.hip:
    # the node part of the edge stage once per node); same prepared buffer, no workspace, no row exponents
    part = getattr(csr, "part", None)
    if (flags.GRAPH_ALIGNED_LAYER and part is not None and d_x == 128 and part[2] <= 128 and int(part[0].numel()) > 1
            and L.gsn_layer_fused_graphs_supported(ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1))):
        n_out_g = node_stages[1].weight.shape[0]
        fl_e = 2.0 * edge_stages[0].weight.shape[1] * edge_stages[0].weight.shape[0]
        fl_n = 2.0 * (node_stages[0].weight.shape[1] * node_stages[0].weight.shape[0] + node_stages[1].weight.shape[1] * n_out_g)
        roles = [None if b[0] is x else i for i, b in enumerate(edge_stages[0].blocks)]      # blocks that are x itself follow the call's x

        def launch_g(x_now, csr_now, blocks_now, y=None):
            """The recorded graph-aligned call with this forward's rows / indices / partition.  None when this batch is outside it."""
            part_now = getattr(csr_now, "part", None)
            if part_now is None or part_now[2] > 128 or int(part_now[0].numel()) <= 1 or x_now.data_ptr() % 16:
                return None
            for i in range(ge.n_blocks):
                d_i, idx_i = blocks_now[i]
                ge.blocks[i].data = d_i.data_ptr()
                ge.blocks[i].idx = None
                ge.blocks[i].idx32 = idx_i.data_ptr()
            n_now, e_now = x_now.shape[0], csr_now.tgt.numel()
            if y is None:
                y = torch.empty((n_now, n_out_g), dtype=torch.float32, device=x_now.device)
            with _abi.device_guard(x_now.device), _timed("layer_fused", fl_e * e_now + fl_n * n_now):
                rcg = L.gsn_layer_fused_fwd_graphs_hip(n_now, e_now, csr_now.seg_ptr.data_ptr(), ctypes.byref(ge), x_now.data_ptr(), d_x, ctypes.byref(g0),
                                                       ctypes.byref(g1), prep.data_ptr(), int(part_now[0].numel()) - 1, part_now[0].data_ptr(), int(part_now[2]),
                                                       y.data_ptr(), _abi.current_stream())
            if rcg == -2:
                return None
            _abi.check(rcg, "gsn_layer_fused_fwd_graphs_hip")
            return y
        launch_g._keep = keep
        got = launch_g(x, csr, edge_stages[0].blocks, out)
        if got is not None:
            if (record is not None and owner is not None
                    and all(b[1] is not None and b[1].dtype == torch.int32 and b[0].dtype == torch.float32 and b[0].is_contiguous() for b in edge_stages[0].blocks)):
                owner._fplan = (record, "graphs", launch_g)
                _note_cache(owner, "_fplan")
            return got
    # layers of a d = 128 model hand the row exponents of their output to the next one (csrc/layer_w.hip takes its edge rows' scales from
    # them): kept on the output tensor together with its version counter, used only while the tensor is unchanged
    x_exp = None
    hit = getattr(x, "_gsn_row_exp", None)
    # (a write through `.data` does not move the version counter -- the caveat of every per-tensor cache here, INTEGRATION.md: the
    #  validation mode does not trust the tensor's exponents and lets the kernel make them again)
    if hit is not None and hit[1] == x._version and hit[0].numel() == n and hit[0].device == x.device and not flags.VALIDATE_CACHES:
        x_exp = hit[0]
    # (asked of the d = 128 kernel only, which writes them with its rows; behind the other kernels they would cost a pass over the output)
    out_exp = torch.empty(n, dtype=torch.int32, device=x.device) if d_x == 128 and out.shape[1] == 128 and flags.CHAIN_ROW_EXPONENTS else None
    ws_bytes = 0 if x_exp is not None else int(L.gsn_layer_fused_workspace_bytes(n, ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1)))
    ws = torch.empty(ws_bytes // 4, dtype=torch.int32, device=x.device) if ws_bytes else None      # (the caching allocator: capture-safe)
    with _abi.device_guard(x.device), _timed("layer_fused", flops):
        rc = L.gsn_layer_fused_fwd_ws_hip(n, E, csr.seg_ptr.data_ptr(), ctypes.byref(ge), x.data_ptr(), d_x, ctypes.byref(g0),
                                          ctypes.byref(g1), prep.data_ptr(), out.data_ptr(), _abi.ptr(ws), ws_bytes, _abi.ptr(x_exp),
                                          _abi.ptr(out_exp), _abi.current_stream())
    if rc == 0 and out_exp is not None:
        out._gsn_row_exp = (out_exp, out._version)
    if rc == -2:           # GSN_E_UNSUPPORTED: this call's arguments are outside the kernel after all (e.g. stream capture on the wide kernel)
        if os.environ.get("GSN_CHAIN_TRACE"):
            import sys
            msg = L.gsn_last_error()
            print("gsn chain: one-launch layer declined: %s" % (msg.decode() if msg else ""), file=sys.stderr)
        return None
    _abi.check(rc, "gsn_layer_fused_fwd_hip")
    return out


def _bn_resolve(stage, stats_fn, m_rows, training, fuse_act=None):
    """Fill stage.bn_params = (mean, scale, shift).  Train mode: batch statistics from a statistics pass (fp64 column
    sums), running statistics updated exactly like nn.BatchNorm1d."""
    bn = stage.bn
    if bn is None:
        stage.bn_params = None
        return
    if training or bn.running_mean is None:
        if training and m_rows == 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size torch.Size([%d, %d])" % (m_rows, bn.num_features))
        if m_rows == 0:
            # no rows (an edge-less batch in front of an edge stage): nn.BatchNorm1d returns the empty tensor, leaves the running statistics
            # alone and still counts the batch; the vectors below are never applied to a row
            n_out, dev = bn.num_features, stage.weight.device
            if training and bn.track_running_stats and bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1             # (an ordinary in-place op: the version counter moves with it eagerly ...)
                if flags.RAW_WRITTEN is not None:              # (... and a replay of the captured add must move it too)
                    flags.RAW_WRITTEN.append(bn.num_batches_tracked)
            vec = torch.zeros((4, n_out), dtype=torch.float32, device=dev)
            vec[1:3] = 1.0
            stage.bn_params = (vec[0], vec[2], vec[3])
            stage.bn_invstd = vec[1]
            return
        stats = stats_fn()
        n_out = stats.shape[1]
        dev = stats.device
        vec = torch.empty((4, n_out), dtype=torch.float32, device=dev)       # mean, invstd, scale, shift
        track = training and bn.track_running_stats and bn.running_mean is not None
        mom = 0.0
        nbt = None
        if track:
            if bn.momentum is not None and bn.num_batches_tracked.is_cuda and bn.num_batches_tracked.dtype == torch.int64:
                mom, nbt = bn.momentum, bn.num_batches_tracked      # (the counter is incremented by the finalize kernel)
            else:
                bn.num_batches_tracked += 1
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        gamma = _f32c(bn.weight) if bn.affine else None
        beta = _f32c(bn.bias) if bn.affine else None
        v0, v1, v2, v3 = vec.unbind(0)           # (the four rows: one call, and their addresses by arithmetic -- this runs per BatchNorm per step)
        p0 = vec.data_ptr()
        with _abi.device_guard(dev):
            if fuse_act is not None:
                # (h, act code, out): the normalise + activate pass rides the same launch (flags.FUSE_BN_ACT_ROWS: where a launch costs more than it)
                fh, fact, fout = fuse_act
                rc = _abi.lib().gsn_bn_finalize_act_hip(n_out, m_rows, float(bn.eps), float(mom), stats.data_ptr(), _abi.ptr(gamma), _abi.ptr(beta),
                                                        bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
                                                        p0, p0 + 4 * n_out, p0 + 8 * n_out, p0 + 12 * n_out, _abi.ptr(nbt),
                                                        fh.data_ptr(), int(fact), fout.data_ptr(), _abi.current_stream())
            else:
                rc = _abi.lib().gsn_bn_finalize_count_hip(n_out, m_rows, float(bn.eps), float(mom), stats.data_ptr(), _abi.ptr(gamma), _abi.ptr(beta),
                                                          bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
                                                          p0, p0 + 4 * n_out, p0 + 8 * n_out, p0 + 12 * n_out,
                                                          _abi.ptr(nbt), _abi.current_stream())
        _abi.check(rc, "gsn_bn_finalize_act_hip" if fuse_act is not None else "gsn_bn_finalize_count_hip")
        if track:
            # the kernel wrote the running statistics (and the counter) through raw pointers: PyTorch's version counters, on which the
            # eval-mode vectors of this module are cached, have to move with them (no launch); a step being captured into a graph notes
            # the tensors so that every REPLAY can do the same (gsn_amd.graphs.GraphedTrainStep)
            touched = [bn.running_mean, bn.running_var] + ([nbt] if nbt is not None else [])
            torch.autograd.graph.increment_version(touched)
            if flags.RAW_WRITTEN is not None:
                flags.RAW_WRITTEN.extend(touched)
        stage.bn_params = (v0, v2, v3)
        stage.bn_invstd = v1
        return
    else:
        # eval mode: the three vectors depend only on the module's buffers / parameters -> cached on their versions
        key = (bn.running_mean._version, bn.running_var._version, bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
               (bn.weight._version, bn.bias._version, bn.weight.data_ptr(), bn.bias.data_ptr()) if bn.affine else None)
        hit = getattr(bn, "_gsn_eval_cache", None)
        if hit is not None and hit[0] == key:
            stage.bn_params = hit[1]
            stage.bn_invstd = hit[2]
            return
        mean32 = bn.running_mean
        invstd = torch.rsqrt(bn.running_var.to(torch.float64) + bn.eps).to(torch.float32)
        scale = invstd * bn.weight.detach() if bn.affine else invstd
        shift = bn.bias.detach() if bn.affine else torch.zeros_like(invstd)
        stage.bn_params = (mean32.contiguous(), scale.contiguous(), shift.contiguous())
        stage.bn_invstd = invstd.contiguous()
        bn._gsn_eval_cache = (key, stage.bn_params, stage.bn_invstd)
        _note_cache(bn, "_gsn_eval_cache")
        return
    scale = invstd * bn.weight.detach() if bn.affine else invstd
    shift = bn.bias.detach() if bn.affine else torch.zeros_like(invstd)
    stage.bn_params = (mean32, scale, shift)


def run_stages(stages, m_rows, training, csr=None):
    """Evaluate a list of _Stage on the HIP kernels.  A train-mode BatchNorm1d stage costs one extra statistics pass over
    the chain prefix that ends at it (the prefix is recomputed, nothing is stored).  ``csr``: fuse the scatter-add
    (returns None, before touching any BatchNorm state, if the stages do not fit one fused launch)."""
    if csr is not None and not _chain_fits(stages):
        return None
    needs_stats = any(st.bn is not None and (training or st.bn.running_mean is None) for st in stages)
    if csr is None and needs_stats and not _chain_fits(stages):
        return _run_stages_materialised(stages, m_rows, training)
    for i, st in enumerate(stages):
        if st.bn is not None:
            def stats_fn(i=i):
                n_out = stages[i].weight.shape[0]
                stats = _zeros(2 * n_out, torch.float64, stages[i].weight.device).view(2, n_out)
                probe = _Stage(stages[i].weight, stages[i].bias, None, "identity", stages[i].blocks)
                _launch_stages(stages[:i] + [probe], m_rows, stats=stats)
                return stats
            _bn_resolve(st, stats_fn, m_rows, training)
    return _launch_stages(stages, m_rows, csr=csr)


def _bn_act_hip(h, bn_params, act):
    """act(bn(h)) in place on materialised pre-BN rows (gsn_bn_act_hip)."""
    vecs = [None if v is None else _f32c(v) for v in (bn_params or (None, None, None))]
    with _abi.device_guard(h.device), _timed("bn_act", 8.0 * h.numel()):
        rc = _abi.lib().gsn_bn_act_hip(h.shape[0], h.shape[1], h.data_ptr(), _abi.ptr(vecs[0]), _abi.ptr(vecs[1]), _abi.ptr(vecs[2]),
                                       _ACT_CODE[act], h.data_ptr(), _abi.current_stream())
    _abi.check(rc, "gsn_bn_act_hip")
    return h


def _run_stages_materialised(stages, m_rows, training):
    """Train-mode stages outside the fused chain (e.g. d = 300): every BatchNorm stage writes its pre-BN rows AND their
    column statistics in ONE pass of the linear kernel, then BatchNorm + activation are applied in place -- instead of a
    statistics pass that recomputes the whole prefix (5 GEMM passes for Linear-BN-act-Linear-BN become 2)."""
    y = None
    for st in stages:
        blks = st.blocks + ([(y, None)] if y is not None else [])
        if st.bn is not None and (training or st.bn.running_mean is None):
            n_out = st.weight.shape[0]
            stats = _zeros(2 * n_out, torch.float64, st.weight.device).view(2, n_out)
            h = _linear_hip(blks, st.weight, st.bias, None, None, None, 0, m_rows, out=True, stats=stats)
            _bn_resolve(st, lambda: stats, m_rows, training)
            y = _bn_act_hip(h, st.bn_params, st.act)
        else:
            _bn_resolve(st, None, m_rows, training)
            bp = st.bn_params or (None, None, None)
            y = _linear_hip(blks, st.weight, st.bias, bp[0], bp[1], bp[2], _ACT_CODE[st.act], m_rows)
    return y

"""Native adjoints of the dense stages composed under autograd (gsn_bn_act_bwd_hip, gsn_wgrad_hip, the forward kernels on W^T), the folded
first weight of update_fn, gathered edge rows, code-gather stages, and the two generic HIP-forward autograd functions."""
from __future__ import annotations

import torch

from . import _abi, flags
from ._caches import _note_cache
from ._dense import _Stage, _bn_resolve, _f16x3_takes, _linear_hip
from ._index import _csr_for, propagate
from ._runtime import _ACT_CODE, _f32c, _timed, _zeros

# ------------------------------------------------------------------------------------------------------------------
# native backward of dense stage lists (inputs = plain row-major blocks): gsn_bn_act_bwd_hip, gsn_wgrad_hip and the
# forward linear kernel on W^T for the input gradient
# ------------------------------------------------------------------------------------------------------------------


def _transposed(w):
    return w.detach().to(torch.float32).t().contiguous()


class _DenseStagesFn(torch.autograd.Function):
    """Forward: stage by stage on the linear kernel, keeping what the adjoint needs (stage outputs; pre-BN rows and batch
    statistics of train-mode BatchNorm stages).  Backward: per stage  gY -> gH (BN + activation adjoint) -> gW, gb, gX."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        # spec: list of dicts {n_blocks, has_bias, bn (module or None), act};  tensors: blocks of stage 0, then per stage
        # weight, [bias], [gamma, beta]
        it = iter(tensors)
        blocks0 = [next(it) for _ in range(spec[0]["n_blocks"])]
        # gathered stage-0 blocks (the x_i / x_j / per-end-point blocks of an edge stage): read where they lie through edge_index[mode],
        # by the forward product AND by the weight gradient -- no assembled [E, K] copy of the rows
        gather = spec[0].get("gather")
        if gather is not None:
            g_ei, g_n, g_modes = gather
            idx0 = [None if m is None else g_ei[m] for m in g_modes]
            m_rows = g_ei.shape[1]
        else:
            idx0 = [None] * len(blocks0)
            m_rows = blocks0[0].shape[0]
        saved, meta = [], []
        y = None
        # per stage: the forward product's row scratch (fp16 planes of the stage's input rows), kept when the weight wants a gradient and the
        # product ran on the fp16x3 kernel -- the backward pass multiplies it with the planes of gH (gsn_wgrad_f16x3_hip)
        x_scratch = []
        presplit_next, y_shape, y_dropped = None, None, set()
        w_pos = spec[0]["n_blocks"]
        for si, sp in enumerate(spec):
            scr = [] if (flags.WGRAD_F16X3 and ctx.needs_input_grad[1 + w_pos] and (si > 0 or gather is None)) else None
            w_pos += 1 + (1 if sp["has_bias"] else 0) + (2 if (sp["bn"] is not None and sp["bn"].affine) else 0)
            # rows split already by the stage before (gsn_bn_act_planes_hip): the product without its pre-pass; `y_shape` stands in for the rows
            pre, presplit_next = presplit_next, None
            w = next(it)
            b = next(it) if sp["has_bias"] else None
            bn = sp["bn"]
            gamma = beta = None
            if bn is not None and bn.affine:
                gamma, beta = next(it), next(it)
            blks = [(t, ix) for t, ix in zip(blocks0, idx0)] if si == 0 else [(y if pre is None else y_shape, None)]
            n_out = w.shape[0]
            bn_train = bn is not None and (bn.training or bn.running_mean is None)
            bn_affine_grad = bn is not None and bn.affine and (bn.weight.requires_grad or bn.bias.requires_grad)
            if bn is not None and not bn_train and not bn_affine_grad:
                # BatchNorm on its running statistics, gamma / beta frozen: a per-column affine map in the epilogue of the product
                st = _Stage(w, b, bn, sp["act"])
                _bn_resolve(st, None, m_rows, False)
                mean32, scale, shift = st.bn_params
                y = _linear_hip(blks, w, b, mean32, scale, shift, _ACT_CODE[sp["act"]], m_rows, scratch_out=scr, presplit=pre)
                saved += [y, _f32c(scale)]
                meta.append(("affine", len(saved) - 2))
            elif bn is not None:
                # batch statistics (train mode), or running statistics with gradients for gamma / beta: pre-BN rows materialised
                st = _Stage(w, b, bn, sp["act"])
                fused_act = False
                if bn_train:
                    stats = _zeros(2 * n_out, torch.float64, w.device).view(2, n_out)
                    h = _linear_hip(blks, w, b, None, None, None, 0, m_rows, out=True, stats=stats, scratch_out=scr, presplit=pre)
                    fused_act = 1 < m_rows <= flags.FUSE_BN_ACT_ROWS
                    yy = torch.empty_like(h) if fused_act else None
                    _bn_resolve(st, lambda: stats, m_rows, True, fuse_act=(h, _ACT_CODE[sp["act"]], yy) if fused_act else None)
                else:
                    h = _linear_hip(blks, w, b, None, None, None, 0, m_rows, out=True, scratch_out=scr, presplit=pre)
                    yy = None
                    _bn_resolve(st, None, m_rows, False)
                mean32, scale, shift = st.bn_params
                vecs = [_f32c(v) for v in (mean32, scale, shift)]
                if not fused_act:
                    # the stage output only feeds the next product (on the fp16x3 kernel) and, in the backward pass, that product's plane weight
                    # gradient: BatchNorm + activation write the row scratch of Y -- no fp32 Y, no pre-pass over it (gsn_bn_act_planes_hip)
                    to_planes = False
                    if flags.BN_ACT_PLANES and si + 1 < len(spec) and n_out % 4 == 0 and n_out <= 640 and h.data_ptr() % 16 == 0 and m_rows > 0:
                        nsp, n_next = spec[si + 1], tensors[w_pos].shape[0]
                        nbn = nsp["bn"]
                        next_stats = nbn is not None and (nbn.training or nbn.running_mean is None)
                        to_planes = (flags.WGRAD_F16X3 and ctx.needs_input_grad[1 + w_pos] and _f16x3_takes(m_rows, n_next, [n_out])
                                     and (not next_stats or (flags.LINEAR_F16X3_STATS and n_next % 4 == 0)))
                    with _abi.device_guard(h.device), _timed("bn_act", 8.0 * h.numel()):
                        if to_planes:
                            presplit_next = torch.empty(int(_abi.lib().gsn_linear_f16x3_scratch_bytes(m_rows, n_out)), dtype=torch.uint8, device=h.device)
                            _abi.check(_abi.lib().gsn_bn_act_planes_hip(m_rows, n_out, h.data_ptr(), vecs[0].data_ptr(), vecs[1].data_ptr(), vecs[2].data_ptr(),
                                                                        _ACT_CODE[sp["act"]], None, presplit_next.data_ptr(), _abi.current_stream()),
                                       "gsn_bn_act_planes_hip")
                            y_shape = h
                            y_dropped.add(si)
                        else:
                            yy = torch.empty_like(h)
                            _abi.check(_abi.lib().gsn_bn_act_hip(m_rows, n_out, h.data_ptr(), vecs[0].data_ptr(), vecs[1].data_ptr(),
                                                                 vecs[2].data_ptr(), _ACT_CODE[sp["act"]], yy.data_ptr(),
                                                                 _abi.current_stream()), "gsn_bn_act_hip")
                invstd = st.bn_invstd
                saved += [h, h if yy is None else yy, vecs[0], invstd.contiguous(), vecs[1], vecs[2]]      # (no fp32 Y: h keeps its slot)
                meta.append(("bn" if bn_train else "bn_eval", len(saved) - 6))
                y = yy
            else:
                y = _linear_hip(blks, w, b, None, None, None, _ACT_CODE[sp["act"]], m_rows, scratch_out=scr, presplit=pre)
                saved += [y]
                meta.append(("plain", len(saved) - 1))
            if pre is not None and scr is not None and not scr:
                scr.append(pre)
            x_scratch.append(scr[0] if scr else None)
        ctx.spec, ctx.meta, ctx.m_rows = spec, meta, m_rows
        ctx.x_scratch = x_scratch
        ctx.y_dropped = y_dropped
        ctx.n_saved = len(saved)
        ctx.save_for_backward(*saved, *tensors)
        return y

    @staticmethod
    def backward(ctx, gy):
        spec, meta, m_rows = ctx.spec, ctx.meta, ctx.m_rows
        allt = ctx.saved_tensors
        saved, tensors = allt[:ctx.n_saved], allt[ctx.n_saved:]
        L = _abi.lib()
        # locate the per-stage tensors again
        pos = spec[0]["n_blocks"]
        blocks0 = list(tensors[:pos])
        per = []
        for sp in spec:
            ent = {"w": tensors[pos], "w_i": pos}
            pos += 1
            if sp["has_bias"]:
                ent["b_i"] = pos; pos += 1
            if sp["bn"] is not None and sp["bn"].affine:
                ent["g"] = tensors[pos]; ent["g_i"] = pos; ent["beta_i"] = pos + 1; pos += 2
            per.append(ent)
        grads = [None] * len(tensors)
        dev = gy.device
        g = gy.to(torch.float32).contiguous()
        # every zero-initialised accumulator of this backward from two arenas (one fill each instead of three small fills per stage)
        n64 = sum(3 * ent["w"].shape[0] for ent in per)
        n32 = sum(ent["w"].numel() for si, ent in enumerate(per) if ctx.needs_input_grad[1 + ent["w_i"]])
        z64 = _zeros(n64, torch.float64, dev)
        z32 = _zeros(n32, torch.float32, dev)
        o64 = o32 = 0
        casts = []
        for si in range(len(spec) - 1, -1, -1):
            sp, ent = spec[si], per[si]
            kind, off = meta[si]
            w = ent["w"]
            n_out, k_total = w.shape
            gbias = z64[o64:o64 + n_out]
            sums_z = z64[o64 + n_out:o64 + 3 * n_out].view(2, n_out)
            o64 += 3 * n_out
            act = _ACT_CODE[sp["act"]]
            # a BatchNorm stage whose gH is read as fp16 planes only -- by the input-gradient product on the fp16x3 kernel (or by nobody) and by the
            # plane weight gradient: the adjoint pass writes gH's row scratch and no fp32 gH (gsn_bn_act_bwd_planes_hip)
            need_x = si > 0 or any(ctx.needs_input_grad[1 + bi] for bi in range(len(blocks0)))
            want_w = ctx.needs_input_grad[1 + ent["w_i"]]
            planes_only = (flags.BN_BWD_PLANES and kind in ("bn", "bn_eval") and want_w and ctx.x_scratch[si] is not None and n_out % 4 == 0 and n_out <= 640
                           and g.data_ptr() % 16 == 0 and (not need_x or _f16x3_takes(m_rows, k_total, [n_out])))
            gh = None if planes_only else torch.empty((m_rows, n_out), dtype=torch.float32, device=dev)
            gh_scratch = None
            with _abi.device_guard(dev), _timed("bn_act_bwd", 16.0 * m_rows * n_out):
                if planes_only:
                    h, y, mean32, invstd, scale, shift = saved[off:off + 6]
                    sums = sums_z
                    gh_scratch = torch.empty(int(L.gsn_linear_f16x3_scratch_bytes(m_rows, n_out)), dtype=torch.uint8, device=dev)
                    rc = L.gsn_bn_act_bwd_planes_hip(m_rows, n_out, g.data_ptr(), h.data_ptr(), mean32.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                                                     shift.data_ptr(), 1 if kind == "bn" else 2, act, sums.data_ptr(), gh_scratch.data_ptr(),
                                                     gbias.data_ptr(), _abi.current_stream())
                elif kind in ("bn", "bn_eval"):
                    h, y, mean32, invstd, scale, shift = saved[off:off + 6]
                    sums = sums_z
                    # (the activation's derivative from z recomputed out of the pre-BN rows: the stage output is not read again)
                    rc = L.gsn_bn_act_bwd_from_h_hip(m_rows, n_out, g.data_ptr(), h.data_ptr(), mean32.data_ptr(), invstd.data_ptr(),
                                                     scale.data_ptr(), shift.data_ptr(), 1 if kind == "bn" else 2, act, sums.data_ptr(),
                                                     gh.data_ptr(), gbias.data_ptr(), _abi.current_stream())
                elif kind == "affine":
                    y, scale = saved[off:off + 2]
                    sums = None
                    rc = L.gsn_bn_act_bwd_hip(m_rows, n_out, g.data_ptr(), y.data_ptr(), None, None, None, scale.data_ptr(), 0, act, None,
                                              gh.data_ptr(), gbias.data_ptr(), _abi.current_stream())
                else:
                    y = saved[off]
                    sums = None
                    rc = L.gsn_bn_act_bwd_hip(m_rows, n_out, g.data_ptr(), y.data_ptr(), None, None, None, None, 0, act, None,
                                              gh.data_ptr(), gbias.data_ptr(), _abi.current_stream())
            _abi.check(rc, "gsn_bn_act_bwd_planes_hip" if planes_only else "gsn_bn_act_bwd_hip")
            # (fp64 column sums -> fp32 gradients: ONE conversion of the whole arena behind the loop, the gradients are its slices)
            o0 = o64 - 3 * n_out
            if kind in ("bn", "bn_eval") and "g_i" in ent:
                casts.append((ent["g_i"], o0 + 2 * n_out, n_out))
                casts.append((ent["beta_i"], o0 + n_out, n_out))
            if "b_i" in ent:
                casts.append((ent["b_i"], o0, n_out))
            # input gradient first: on the fp16x3 kernel its row pre-pass leaves the fp16 planes of gH, which the weight gradient multiplies as well
            xs = ctx.x_scratch[si] if want_w else None
            g_scr = [] if xs is not None else None
            gx = None
            if need_x:
                # gX = gH W: W read as its transpose.  The fp16x3 kernel prepares its planes from any strides; the bf16x6 kernel stages a
                # strided W with scalar loads -- fine where a launch costs more than the staging (small batches), a copy + float4 staging above
                wt = w.detach().t() if (w.shape[1] > flags.LINEAR_F16X3_MIN_N or m_rows <= 16384) else _transposed(w)
                if planes_only:         # (the rows are split already; `h` stands in for gH's shape)
                    gx = _linear_hip([(saved[off], None)], wt, None, None, None, None, 0, m_rows, split_k=True, presplit=gh_scratch)
                else:
                    gx = _linear_hip([(gh, None)], wt, None, None, None, None, 0, m_rows, split_k=True, scratch_out=g_scr)
            if planes_only:
                g_scr.append(gh_scratch)
            # weight gradient
            def xin_rows():
                if si == 0:
                    return blocks0
                poff = meta[si - 1][1]
                if (si - 1) in ctx.y_dropped:      # (the stage output was written as planes only: the fp32 rows again, from the pre-BN rows)
                    ph, _, pmean, _, pscale, pshift = saved[poff:poff + 6]
                    yy = torch.empty_like(ph)
                    with _abi.device_guard(dev):
                        _abi.check(L.gsn_bn_act_hip(m_rows, ph.shape[1], ph.data_ptr(), pmean.data_ptr(), pscale.data_ptr(), pshift.data_ptr(),
                                                    _ACT_CODE[spec[si - 1]["act"]], yy.data_ptr(), _abi.current_stream()), "gsn_bn_act_hip")
                    return [yy]
                return [saved[poff + (1 if meta[si - 1][0] in ("bn", "bn_eval") else 0)]]
            if want_w:
                gw = z32[o32:o32 + n_out * k_total].view(n_out, k_total)
                o32 += n_out * k_total
                if xs is not None and not g_scr and n_out % 4 == 0 and gh is not None and gh.data_ptr() % 16 == 0 and m_rows >= flags.WGRAD_F16X3_SPLIT_ROWS:
                    # no input-gradient product on the fp16x3 kernel beside it: the planes of gH from the pre-pass alone
                    sc = torch.empty(int(L.gsn_linear_f16x3_scratch_bytes(m_rows, n_out)), dtype=torch.uint8, device=dev)
                    one = (_abi.gsn_block * 1)()
                    one[0].data = gh.data_ptr(); one[0].idx = None; one[0].idx32 = None; one[0].width = n_out
                    with _abi.device_guard(dev):
                        _abi.check(L.gsn_linear_f16x3_split_rows_hip(m_rows, 1, one, sc.data_ptr(), _abi.current_stream()), "gsn_linear_f16x3_split_rows_hip")
                    g_scr.append(sc)
                if xs is not None and g_scr:
                    with _abi.device_guard(dev), _timed("wgrad", 2.0 * m_rows * n_out * k_total):
                        _abi.check(L.gsn_wgrad_f16x3_hip(m_rows, n_out, k_total, g_scr[0].data_ptr(), xs.data_ptr(), gw.data_ptr(), _abi.current_stream()),
                                   "gsn_wgrad_f16x3_hip")
                else:
                    xin = xin_rows()
                    arr = (_abi.gsn_block * len(xin))()
                    keep = []
                    gather = spec[0].get("gather") if si == 0 else None
                    for bi, t in enumerate(xin):
                        t = _f32c(t); keep.append(t)
                        arr[bi].data = t.data_ptr(); arr[bi].idx = None; arr[bi].idx32 = None; arr[bi].width = t.shape[1]
                        if gather is not None and gather[2][bi] is not None:
                            ix = gather[0][gather[2][bi]].contiguous(); keep.append(ix)
                            arr[bi].idx = ix.data_ptr()
                    with _abi.device_guard(dev), _timed("wgrad", 2.0 * m_rows * n_out * k_total):
                        _abi.check(L.gsn_wgrad_hip(m_rows, n_out, gh.data_ptr(), len(xin), arr, gw.data_ptr(), _abi.current_stream()),
                                   "gsn_wgrad_hip")
                grads[ent["w_i"]] = gw
            if need_x:
                if si > 0:
                    g = gx
                else:
                    gather = spec[0].get("gather")
                    o = 0
                    for bi, t in enumerate(blocks0):
                        wd = t.shape[1]
                        if ctx.needs_input_grad[1 + bi]:
                            mode = None if gather is None else gather[2][bi]
                            if mode is None:
                                grads[bi] = gx[:, o:o + wd]
                            else:       # rows gathered through edge_index[mode]: the per-edge gradients summed per vertex (the propagate kernel)
                                grads[bi] = _segment_sum_cols(gather[0], mode, gather[1], gx, o, wd)
                        o += wd
        if casts:
            c32 = z64.to(torch.float32)
            for gi, o0, n in casts:
                grads[gi] = c32[o0:o0 + n]
        return (None,) + tuple(grads)


def _segment_sum_cols(edge_index, mode, n_nodes, rows, col0, width):
    """sum over the columns e of edge_index with edge_index[mode, e] = v of rows[e, col0 : col0 + width] -> [n_nodes, width]: the input gradient
    of a block gathered through edge_index[mode], read where the input-gradient product left it (gsn_segment_sum_rows_hip: the slice is not
    copied).  Slices that are not 16-byte aligned take the copy + propagate route."""
    E = rows.shape[0]
    if E == 0 or rows.dtype is not torch.float32 or rows.stride(1) != 1 or (col0 | width | rows.stride(0)) % 4 or rows.data_ptr() % 16:
        with torch.no_grad():
            return propagate(0, edge_index, mode, n_nodes, b=rows[:, col0:col0 + width].contiguous())
    csr = _csr_for(edge_index, mode, n_nodes)
    src = edge_index[1 - mode].contiguous()
    out = torch.empty((n_nodes, width), dtype=torch.float32, device=rows.device)
    with _abi.device_guard(rows.device), _timed("propagate_fwd", 12.0 * E + 4.0 * n_nodes + 4.0 * (E + n_nodes) * width):
        rc = _abi.lib().gsn_segment_sum_rows_hip(n_nodes, E, src.data_ptr(), csr.seg_ptr.data_ptr(), csr.perm.data_ptr(),
                                                 csr.src.data_ptr() if csr.src is not None else None, rows.data_ptr() + 4 * col0, width,
                                                 rows.stride(0), out.data_ptr(), _abi.current_stream())
    _abi.check(rc, "gsn_segment_sum_rows_hip")
    return out


class _FoldWeightsFn(torch.autograd.Function):
    """w_first = [W3[:, :d_x] | W3[:, d_x:] W2 | W3[:, d_x:] b2 | pad zero columns]: update_fn's first weight with msg_fn's last Linear (W2, b2) folded in
    (GSN_edge_sparse.py:153-170: update_fn(cat(x, sum_e msg_fn(...)))), differentiable in W3, W2 and b2; one launch each way."""

    @staticmethod
    def takes(w3, last, d_x):
        w2, b2 = last.weight, last.bias
        return (b2 is not None and w3.is_cuda and all(t.dtype is torch.float32 and t.is_contiguous() for t in (w3, w2, b2))
                and w3.shape[1] - d_x == w2.shape[0] and max(w3.shape[0], w2.shape[0], w2.shape[1]) <= 8192
                and w3.shape[0] * w2.shape[0] * w2.shape[1] <= (1 << 25))      # (plain FMA dot products: the matrices of a layer, not a workload)

    @staticmethod
    def forward(ctx, w3, w2, b2, d_x, pad=0):
        R, A, H = w3.shape[0], w2.shape[0], w2.shape[1]
        out = torch.empty((R, d_x + H + 1 + pad), dtype=torch.float32, device=w3.device)
        with _abi.device_guard(w3.device):
            rc = _abi.lib().gsn_fold_weights_fwd_hip(R, d_x, A, H, pad, w3.data_ptr(), w3.stride(0), w2.data_ptr(), w2.stride(0), b2.data_ptr(),
                                                     out.data_ptr(), _abi.current_stream())
        _abi.check(rc, "gsn_fold_weights_fwd_hip")
        ctx.save_for_backward(w3, w2, b2)
        ctx.d_x = d_x
        return out

    @staticmethod
    def backward(ctx, g):
        w3, w2, b2 = ctx.saved_tensors
        d_x, R, A, H = ctx.d_x, w3.shape[0], w2.shape[0], w2.shape[1]
        if g.dtype is not torch.float32 or g.stride(1) != 1:
            g = g.to(torch.float32).contiguous()
        g_w3, g_w2, g_b2 = torch.empty_like(w3), torch.empty_like(w2), torch.empty_like(b2)
        with _abi.device_guard(w3.device):
            rc = _abi.lib().gsn_fold_weights_bwd_hip(R, d_x, A, H, g.data_ptr(), g.stride(0), w3.data_ptr(), w3.stride(0), w2.data_ptr(), w2.stride(0),
                                                     b2.data_ptr(), g_w3.data_ptr(), g_w2.data_ptr(), g_b2.data_ptr(), _abi.current_stream())
        _abi.check(rc, "gsn_fold_weights_bwd_hip")
        return g_w3, g_w2, g_b2, None, None


def _dense_native_ok(stages, training=None):
    """Native backward covers: plain (un-gathered) blocks, <= 5 of them; BatchNorm on batch statistics or (r03) on its running
    statistics -- each BatchNorm1d module's own ``training`` flag decides, as in the reference (models_misc.py:41-45)."""
    if not flags.NATIVE_DENSE_BACKWARD or not stages or len(stages[0].blocks) > 5:
        return False
    for i, st in enumerate(stages):
        if any(idx is not None for _, idx in st.blocks) or (i > 0 and st.blocks):
            return False
    return True


def run_stages_autograd(stages, m_rows, training, gather=None):
    """Differentiable evaluation of a dense stage list with the native adjoint.  ``gather = (edge_index, n_nodes, modes)``: block b of
    the first stage is gathered through ``edge_index[modes[b]]`` (None: one row per edge) -- the edge rows cat(x_i, x_j, ..) of
    GSN_sparse.py:166-171 are then never assembled, neither for the product nor for the weight gradient."""
    spec, tensors = [], []
    tensors += [d for d, _ in stages[0].blocks]
    for i, st in enumerate(stages):
        spec.append({"n_blocks": len(st.blocks) if i == 0 else 0, "has_bias": st.bias is not None, "bn": st.bn, "act": st.act})
        if i == 0 and gather is not None and any(m is not None for m in gather[2]):
            spec[0]["gather"] = gather
        tensors.append(st.weight)
        if st.bias is not None:
            tensors.append(st.bias)
        if st.bn is not None and st.bn.affine:
            tensors += [st.bn.weight, st.bn.bias]
    return _DenseStagesFn.apply(spec, *tensors)


def _transposed_weight(lin):
    key = (lin.weight._version, lin.weight.data_ptr())
    hit = getattr(lin, "_gsn_wt", None)
    if hit is None or hit[0] != key:
        hit = (key, lin.weight.detach().to(torch.float32).t().contiguous())
        lin._gsn_wt = hit
        _note_cache(lin, "_gsn_wt")
    return hit[1]


def _code_stage_segsum(mf, cblocks, csr, m_rows):
    """msg_fn's first stage over Codes blocks + sum per target: gsn_code_stage_fwd_hip.  ``cblocks``: (Codes, int32 row
    index per target-sorted position) in concatenation order.  Returns [n_nodes, d_h] or None if the shape does not fit."""
    L = _abi.lib()
    lin = mf.fc[0]
    bn = mf.bn[0] if mf.batch_norm else None
    n_out, k_total = lin.weight.shape
    n_slots = sum(len(c.n_classes) for c, _ in cblocks)
    if n_slots > 16 or sum(sum(c.n_classes) for c, _ in cblocks) != k_total or not L.gsn_code_stage_supported(n_slots, k_total, n_out):
        return None
    dev = lin.weight.device
    arr = (_abi.gsn_code_slot * n_slots)()
    s, off = 0, 0
    for c, idx in cblocks:
        for col, ncls in enumerate(c.n_classes):
            arr[s].codes = c.codes.data_ptr(); arr[s].idx = idx.data_ptr()
            arr[s].stride = c.codes.shape[1]; arr[s].col = col; arr[s].w_off = off; arr[s].n_classes = ncls
            arr[s].clamp = int(c.clamp)
            s += 1
            off += ncls
    wt = _transposed_weight(lin)
    bias = _f32c(lin.bias)
    status = _zeros(1, torch.int32, dev)

    def launch(bn_params, out, stats):
        vecs = [None if v is None else _f32c(v) for v in (bn_params or (None, None, None))]
        with _abi.device_guard(dev), _timed("code_stage", 4.0 * m_rows * n_slots * n_out):
            rc = L.gsn_code_stage_fwd_hip(m_rows, n_slots, arr, wt.data_ptr(), k_total, bias.data_ptr(), n_out,
                                          _abi.ptr(vecs[0]), _abi.ptr(vecs[1]), _abi.ptr(vecs[2]), _ACT_CODE[mf.activation_name],
                                          csr.tgt.data_ptr(), _abi.ptr(out), _abi.ptr(stats), status.data_ptr(),
                                          _abi.current_stream())
        _abi.check(rc, "gsn_code_stage_fwd_hip")

    stage = _Stage(lin.weight, lin.bias, bn, mf.activation_name)

    def stats_fn():
        stats = _zeros(2 * n_out, torch.float64, dev).view(2, n_out)
        launch(None, None, stats)
        return stats

    _bn_resolve(stage, stats_fn, m_rows, mf.training)
    n_seg = csr.seg_ptr.numel() - 1
    out = torch.empty((n_seg, n_out), dtype=torch.float32, device=dev)
    with _abi.device_guard(dev), _timed("segsum_prepare"):
        _abi.check(L.gsn_segsum_prepare_hip(n_seg, m_rows, csr.seg_ptr.data_ptr(), csr.tgt.data_ptr(), n_out, out.data_ptr(),
                                            _abi.current_stream()), "gsn_segsum_prepare_hip")
    launch(stage.bn_params, out, None)
    if flags.CODE_STATUS_CHECK and int(status.item()) != 0:
        raise IndexError("a code is outside [0, n_classes) of its column")
    return out


class _GatherCatFn(torch.autograd.Function):
    """cat(x[idx_i], x[idx_j], ids.., e) for the training path (gsn_gather_cat_hip); the adjoint of a gathered block is the
    scatter-add over its index = the propagate kernel on the cached CSR of that edge_index row."""

    @staticmethod
    def forward(ctx, edge_index, n_nodes, modes, *tensors):
        # modes[b]: 0 / 1 = rows gathered through edge_index[0] / edge_index[1], None = one row per edge
        E = edge_index.shape[1]
        ts = [_f32c(t) for t in tensors]
        arr = (_abi.gsn_block * len(ts))()
        keep = []
        for b, (t, m) in enumerate(zip(ts, modes)):
            arr[b].data = t.data_ptr(); arr[b].width = t.shape[1]; arr[b].idx32 = None; arr[b].idx = None
            if m is not None:
                idx = edge_index[m].contiguous(); keep.append(idx)
                arr[b].idx = idx.data_ptr()
        k_total = sum(t.shape[1] for t in ts)
        out = torch.empty((E, k_total), dtype=torch.float32, device=edge_index.device)
        with _abi.device_guard(out.device), _timed("gather_cat", 8.0 * out.numel()):
            _abi.check(_abi.lib().gsn_gather_cat_hip(E, len(ts), arr, out.data_ptr() if E else None, _abi.current_stream()),
                       "gsn_gather_cat_hip")
        ctx.edge_index, ctx.n_nodes, ctx.modes = edge_index, n_nodes, modes
        ctx.widths = [t.shape[1] for t in ts]
        return out

    @staticmethod
    def backward(ctx, g):
        grads, o = [], 0
        for b, (w, m) in enumerate(zip(ctx.widths, ctx.modes)):
            if not ctx.needs_input_grad[3 + b]:
                grads.append(None)
            else:
                gb = g[:, o:o + w].contiguous()
                # rows gathered through edge_index[m]: sum the per-edge gradients per vertex of that row
                grads.append(gb if m is None else propagate(0, ctx.edge_index, m, ctx.n_nodes, b=gb))
            o += w
        return (None, None, None) + tuple(grads)


class _HipWithTorchBackward(torch.autograd.Function):
    """y = hip_fn() in forward; gradients by re-running a differentiable evaluation of the same function under autograd: the PyTorch
    twin here, a composition of kernels that each have a HIP adjoint in the subclass below (same mechanics)."""

    @staticmethod
    def forward(ctx, hip_fn, torch_fn, n_inputs, *tensors):
        ctx.torch_fn, ctx.n_inputs = torch_fn, n_inputs
        ctx.params = list(tensors[n_inputs:])   # the module's own Parameter objects (the twin reads them directly)
        ctx.save_for_backward(*tensors[:n_inputs])
        with torch.no_grad():
            return hip_fn()

    @staticmethod
    def backward(ctx, gy):
        tensors = ctx.saved_tensors
        ins = [t.detach().requires_grad_(ctx.needs_input_grad[3 + i]) for i, t in enumerate(tensors)]
        params = ctx.params
        with torch.enable_grad():
            y = ctx.torch_fn(*ins)
            wanted = [t for t in ins if t.requires_grad] + [p for i, p in enumerate(params) if ctx.needs_input_grad[3 + ctx.n_inputs + i]]
            grads = torch.autograd.grad(y, wanted, gy, allow_unused=True) if wanted else []
        it = iter(grads)
        out = [None, None, None]
        for t in ins:
            out.append(next(it) if t.requires_grad else None)
        for i, p in enumerate(params):
            out.append(next(it) if ctx.needs_input_grad[3 + ctx.n_inputs + i] else None)
        return tuple(out)


class _HipWithNativeBackward(_HipWithTorchBackward):
    """Same, with ``torch_fn`` a composition of HIP kernels with HIP adjoints (eval-mode gradients: the fused forward keeps nothing,
    the backward re-runs the stages materialised and walks their adjoints -- gsn_bn_act_bwd_hip, gsn_wgrad_hip, gsn_propagate_bwd_hip)."""


def _run(module, hip_fn, torch_fn, inputs, extra_params=(), native=False):
    if not torch.is_grad_enabled():
        return hip_fn()
    params = [p for p in module.parameters()] + list(extra_params)
    need_grad = torch.is_grad_enabled() and (any(t.requires_grad for t in inputs) or any(p.requires_grad for p in params))
    if not need_grad:
        with torch.no_grad():
            return hip_fn()
    fn = _HipWithNativeBackward if (native and flags.NATIVE_DENSE_BACKWARD) else _HipWithTorchBackward
    return fn.apply(hip_fn, torch_fn, len(inputs), *inputs, *params)

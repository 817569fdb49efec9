"""HP-1 + HP-2 of a collated batch as ONE host call and TWO kernel launches (``gsn_count_layer_step_hip``).

What the reference reaches in three Python-level stages -- ``subgraph_counts2ids`` (utils_ids.py:7-29: the int64 identifiers), the
one-hot encoders on ``data.x`` / ``data.edge_features`` / ``data.identifiers`` (utils_graph_learning.py:170-187, called from
models_graph_classification.py:204-222) and ``GSN_edge_sparse.forward`` of layer 0 (GSN_edge_sparse.py:82-170, which re-sorts the
edges into a sparse tensor, :136-139) -- runs here as

* the counting kernel, whose workgroups also leave the target-sorted CSR of their graph, the node pack (one-hot of the atom codes) and
  the whole edge pack rows (identifier classes + one-hot of the bond codes): ``gsn_count_encode_pack16_side_hip``;
* the one-launch layer on those packs: ``gsn_layer_fused_fwd_pack16_hip``.

``CountLayerStep`` keeps the two argument structs of the C entry filled in, so a step costs one foreign call (the eager composition
``count_batch`` + ``layer(Codes, ...)`` costs six launches through ~0.3 ms of Python).  Results are those of the composition, bit for bit
(``tests/test_step_gpu.py``).  There is no fallback: a layer / plan the two kernels do not take raises at construction.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _abi, _dense, flags, packs
from ._index import Codes, _CSR


class CountLayerStep:
    """``step = CountLayerStep(plan, layer, id_classes)``; ``ids, y, status = step(node_ptr, edge_ptr, edge_index, x_codes, ef_codes,
    max_nodes, max_edges)``.

    plan        edge-mode :class:`gsn_amd.counting.CountPlan` (GSN-e identifiers)
    layer       a ``GSN_edge_sparse`` with msg_kind='general', id_scope='local', two-stage msg_fn, every width 128, in eval mode
                (the shapes of ``gsn_layer_fused_pack16_supported``)
    id_classes  classes per identifier column of the one-hot encoder (counts above the last class are clamped to it when ``clamp``)
    x_codes / ef_codes   :class:`gsn_amd.layers.Codes` (int64 codes + class counts) of the batch's vertices / columns
    Returns the int64 identifiers [E, plan.n_cols], the layer output fp32 [N, d_out] and the per-graph status words (device, not read)."""

    def __init__(self, plan, layer, id_classes, clamp=True):
        if plan.mode != "edge":
            raise ValueError("CountLayerStep: an edge-mode plan (GSN-e identifiers)")
        self.plan, self.layer = plan, layer
        self.id_classes = [int(c) for c in id_classes]
        if len(self.id_classes) != plan.n_cols or min(self.id_classes) < 1:
            raise ValueError("CountLayerStep: one class count >= 1 per identifier column (%d columns)" % plan.n_cols)
        self.clamp = bool(clamp)
        self._enc_tab = np.asarray(self.id_classes, dtype=np.int32)
        if (layer.ogb or layer.msg_kind != "general" or len(layer.msg_fn.fc) != 2 or not layer.has_ids or layer.id_scope != "local"):
            raise ValueError("CountLayerStep: a `general` GSN_edge_sparse layer with id_scope='local' and a two-stage msg_fn")
        self._bufs = None          # (key, dict) of the batch-shaped device buffers
        self._lay = None           # (key, structs) of the layer call
        self._count_call = _abi.gsn_count_call()
        self._side = _abi.gsn_count_side()
        self._layer_call = _abi.gsn_layer_pack16_call()
        self._pk = _abi.gsn_pack16()
        self._bound = None

    # ---- buffers and structs ----------------------------------------------------------------------------------------------
    def _buffers(self, N, E, G, dev):
        key = (N, E, G, str(dev))
        if self._bufs is None or self._bufs[0] != key:
            b = {
                "seg_ptr": torch.empty(N + 1, dtype=torch.int32, device=dev), "perm": torch.empty(max(E, 1), dtype=torch.int32, device=dev),
                "tgt": torch.empty(max(E, 1), dtype=torch.int32, device=dev), "src": torch.empty(max(E, 1), dtype=torch.int32, device=dev),
                # (every row of both packs is written whole by the counting workgroups: no zero fill)
                "npack": torch.empty((N, packs.NODE_COLS), dtype=torch.float16, device=dev),
                "epack": torch.empty((max(E, 1), packs.EDGE_COLS), dtype=torch.float16, device=dev),
                "status": torch.empty(max(G, 1), dtype=torch.int32, device=dev), "code_status": torch.zeros(1, dtype=torch.int32, device=dev),
            }
            self._bufs = (key, b)
            self._lay = None
            self._bound = None
        return self._bufs[1]

    def _layer_structs(self, b, N, E, d_x, w_e):
        """The stage descriptors / prepared weights of the layer call (rebuilt when a parameter or a BatchNorm buffer moved)."""
        layer = self.layer
        if layer.training:
            raise RuntimeError("CountLayerStep: the layer must be in eval mode (train-mode BatchNorm takes batch statistics: not this kernel)")
        mf, uf = layer.msg_fn, layer.update_fn
        w_first = layer._folded_first_weight(d_x)
        dev = b["npack"].device
        sb = [(torch.empty((0, d_x), dtype=torch.float32, device=dev), None)]   # (only for mlp.stages' width bookkeeping)
        edge_stages = mf.stages(sb, upto=len(mf.fc) - 1)
        node_stages = uf.stages(sb, first_weight=w_first, post=None)
        stages = edge_stages + node_stages
        for st in stages:
            if st.act not in ("identity", "relu"):
                raise ValueError("CountLayerStep: activation %r is outside the one-launch layer kernel" % st.act)
            if st.bn is not None and (st.bn.training or st.bn.running_mean is None):
                raise RuntimeError("CountLayerStep: a BatchNorm1d in train mode / without running statistics")
        key = (tuple(_dense._prep_key(st) for st in stages), d_x, getattr(layer, "_fold_gen", 0), N, E)
        if self._lay is not None and self._lay[0] == key:
            return self._lay[1]
        for st in stages:
            _dense._bn_resolve(st, None, 0, False)
        keep = []
        ge = _dense._stage_struct(edge_stages[0], [], keep)
        # the edge stage's blocks: x through the sorted targets, x through the sorted sources, the edge pack's columns through perm
        barr = (_abi.gsn_block * 3)()
        x_ptr = b["npack"].data_ptr()
        for i, (ptr, width, idx) in enumerate(((x_ptr, d_x, b["tgt"]), (x_ptr, d_x, b["src"]), (b["epack"].data_ptr(), w_e, b["perm"]))):
            barr[i].data = ptr; barr[i].width = width; barr[i].idx = None; barr[i].idx32 = idx.data_ptr()
        keep.append(barr)
        ge.blocks = barr; ge.n_blocks = 3
        g0 = _dense._stage_struct(node_stages[0], [], keep)
        g1 = _dense._stage_struct(node_stages[1], [], keep)
        L = _abi.lib()
        if node_stages[0].weight.shape[1] != d_x + edge_stages[0].weight.shape[0] + 4 or edge_stages[0].weight.shape[1] != 2 * d_x + w_e:
            raise ValueError("CountLayerStep: the layer's widths do not match the codes (d_x %d, edge-level columns %d)" % (d_x, w_e))
        if not L.gsn_layer_fused_pack16_supported(ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1)):
            raise ValueError("CountLayerStep: shape outside the packed-row layer kernel (every stage 128 wide, d_x + 4 <= 32, <= 16 edge-level columns)")
        nbytes = int(L.gsn_layer_fused_pack16_prepared_bytes(ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1)))
        prep = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
        with _abi.device_guard(dev):
            _abi.check(L.gsn_layer_fused_pack16_prepare_hip(ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1), prep.data_ptr(),
                                                            _abi.current_stream()), "gsn_layer_fused_pack16_prepare_hip")
        d_out = node_stages[1].weight.shape[0]
        flops = 2.0 * E * edge_stages[0].weight.shape[1] * edge_stages[0].weight.shape[0]
        flops += 2.0 * N * (node_stages[0].weight.shape[1] * node_stages[0].weight.shape[0] + node_stages[1].weight.shape[1] * d_out)
        val = (ge, g0, g1, prep, keep, d_out, flops)
        self._lay = (key, val)
        self._bound = None
        return val

    # ---- the step ---------------------------------------------------------------------------------------------------------
    def __call__(self, node_ptr, edge_ptr, edge_index, x_codes, ef_codes, max_nodes, max_edges, ids_out=None, ids_are_global=True, out=None):
        _abi.require_gpu()
        if not (isinstance(x_codes, Codes) and isinstance(ef_codes, Codes)):
            raise TypeError("CountLayerStep: x_codes / ef_codes are gsn_amd.layers.Codes (integer codes + class counts)")
        dev = edge_index.device
        N, E, G = x_codes.codes.shape[0], edge_index.shape[1], node_ptr.numel() - 1
        if ef_codes.codes.shape[0] != E or edge_index.dtype != torch.int64 or edge_index.shape[0] != 2 or edge_index.stride(1) != 1:
            raise ValueError("CountLayerStep: edge_index int64 [2, E] with unit column stride, one edge code row per column")
        if E == 0 or G == 0:
            raise ValueError("CountLayerStep: an edge-less batch has no GSN-e identifiers (use the layer's own forward)")
        d_x, w_ids, w_ef = sum(x_codes.n_classes), sum(self.id_classes), sum(ef_codes.n_classes)
        if d_x > packs.NODE_COLS - 4 or w_ids + w_ef > packs.EDGE_COLS or len(x_codes.n_classes) > 4 or len(ef_codes.n_classes) > 4 or w_ids % 4 or w_ef > 8:
            raise ValueError("CountLayerStep: code widths outside the packs (node %d <= 28, edge %d + %d <= 16, identifier classes a multiple of 4, "
                             "<= 8 edge code classes)" % (d_x, w_ids, w_ef))
        b = self._buffers(N, E, G, dev)
        ge, g0, g1, prep, _keep, d_out, flops = self._layer_structs(b, N, E, d_x, w_ids + w_ef)
        if ids_out is None:
            ids_out = torch.empty((E, self.plan.n_cols), dtype=torch.int64, device=dev)
        y = out if out is not None else torch.empty((N, d_out), dtype=torch.float32, device=dev)
        c, s, l = self._count_call, self._side, self._layer_call
        bound = (node_ptr.data_ptr(), edge_ptr.data_ptr(), edge_index.data_ptr(), edge_index.stride(0), x_codes.codes.data_ptr(), ef_codes.codes.data_ptr(),
                 int(max_nodes), int(max_edges), bool(ids_are_global), tuple(x_codes.n_classes), tuple(ef_codes.n_classes), x_codes.clamp, ef_codes.clamp, id(prep))
        if self._bound != bound:
            tab = self.plan.device_table(dev)
            c.plan_host = _abi.ptr(self.plan.table); c.plan_dev = tab.data_ptr(); c.plan_words = len(self.plan.table); c.n_graphs = G
            c.node_ptr = node_ptr.data_ptr(); c.edge_ptr = edge_ptr.data_ptr(); c.edge_index = edge_index.data_ptr(); c.edge_row_stride = edge_index.stride(0)
            c.ids_are_global = int(bool(ids_are_global)); c.max_nodes = int(max_nodes); c.max_edges = int(max_edges)
            c.status = b["status"].data_ptr(); c.n_classes = _abi.ptr(self._enc_tab); c.clamp = int(self.clamp)
            c.pack = b["epack"].data_ptr(); c.pack_stride = packs.EDGE_COLS; c.pack_col0 = 0
            s.csr_row = self.layer._sel(); s.seg_ptr = b["seg_ptr"].data_ptr(); s.perm = b["perm"].data_ptr()
            s.sorted_target = b["tgt"].data_ptr(); s.sorted_other = b["src"].data_ptr(); s.n_nodes = N; s.n_edges = E
            s.node_codes = x_codes.codes.data_ptr(); s.node_code_cols = len(x_codes.n_classes); s.node_clamp = int(x_codes.clamp)
            s.edge_codes = ef_codes.codes.data_ptr(); s.edge_code_cols = len(ef_codes.n_classes); s.edge_clamp = int(ef_codes.clamp)
            for i in range(4):
                s.node_n_classes[i] = x_codes.n_classes[i] if i < len(x_codes.n_classes) else 0
                s.edge_n_classes[i] = ef_codes.n_classes[i] if i < len(ef_codes.n_classes) else 0
            s.node_pack = b["npack"].data_ptr(); s.edge_col0 = w_ids; s.code_status = b["code_status"].data_ptr()
            c.side = ctypes.pointer(s)
            self._pk.node_rows = b["npack"].data_ptr(); self._pk.edge_rows = b["epack"].data_ptr()
            l.n_nodes = N; l.n_edges = E; l.seg_ptr = b["seg_ptr"].data_ptr(); l.edge = ctypes.pointer(ge); l.x = b["npack"].data_ptr(); l.d_x = d_x
            l.node0 = ctypes.pointer(g0); l.node1 = ctypes.pointer(g1); l.prepared = prep.data_ptr(); l.pack = ctypes.pointer(self._pk); l.edge_rows = E
            self._bound = bound
        c.out = ids_out.data_ptr()
        l.out = y.data_ptr()
        timer = flags.KERNEL_TIMER
        if timer is None:
            with _abi.device_guard(dev):
                rc = _abi.lib().gsn_count_layer_step_hip(ctypes.byref(c), ctypes.byref(l), None, _abi.current_stream())
        else:
            # measuring host (bench.py): HIP events around the two kernels of the one call -- the middle one is recorded by the library between its
            # two launches; entries under the names the separate launches use ("count", "layer_fused")
            only = flags.KERNEL_TIMER_ONLY
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            with _abi.device_guard(dev):
                ev[1].record()                         # (creates the handle the library records again below)
                if only is None or "count" in only:
                    ev[0].record()
                rc = _abi.lib().gsn_count_layer_step_hip(ctypes.byref(c), ctypes.byref(l), ev[1].cuda_event, _abi.current_stream())
                ev[2].record()
            if only is None or "count" in only:
                # bytes of the counting launch: edge_index, int64 identifiers, the identifier columns of the pack; side workgroups: codes in, node pack,
                # edge-code columns, CSR arrays out
                timer.setdefault("count", []).append((ev[0], ev[1], 16.0 * E + 8.0 * E * self.plan.n_cols + 2.0 * E * w_ids + 16.0 * E
                                                      + 72.0 * N + 16.0 * E + 12.0 * E + 4.0 * N))
            if only is None or "layer_fused" in only:
                timer.setdefault("layer_fused", []).append((ev[1], ev[2], flops))
        _abi.check(rc, "gsn_count_layer_step_hip")
        return ids_out, y, b["status"]

    def csr(self):
        """The CSR arrays the last step wrote (seg_ptr, perm, sorted targets, sorted sources) as a layers-side _CSR object."""
        b = self._bufs[1]
        c = _CSR()
        c.seg_ptr, c.perm, c.tgt, c.src, c.part = b["seg_ptr"], b["perm"], b["tgt"], b["src"], None
        c._deg = c._deg4 = None
        return c

    def packs(self):
        """(node pack, edge pack) the last step wrote."""
        b = self._bufs[1]
        return b["npack"], b["epack"]

    def check_status(self):
        """Read the status words of the last step back (a host synchronisation) and raise the reference's errors."""
        from .counting import _raise_statuses
        b = self._bufs[1]
        _raise_statuses(b["status"])
        if int(b["code_status"].item()) != 0:
            raise IndexError("a code outside its column's classes (one-hot encoding of integer codes)")

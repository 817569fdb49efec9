"""Sparse GSN / MPNN message-passing layers on HIP kernels: host-side mirror of the reference's ``graph_filters``.

Classes, constructor arguments, ``forward(x, edge_index, **kwargs)`` signature, error behaviour and state_dict key
names follow graph_filters/GSN_sparse.py, GSN_edge_sparse.py, MPNN_sparse.py, MPNN_edge_sparse.py,
GSN_edge_sparse_ogb.py, MPNN_edge_sparse_ogb.py and models_misc.mlp of the reference, so reference checkpoints load
(`load_state_dict`) and ``models_graph_classification.py`` can instantiate them unchanged.

Forward pass = libgsn_hip.so kernels only:
  * gsn_csr_build_hip        target-sorted CSR of the batch (cached per edge_index)
  * gsn_mlp_chain_fwd_hip    one or two fused Linear(+BatchNorm1d)(+activation) stages, input rows gathered / concatenated on
                             the fly, optionally with the scatter-add fused into the epilogue (gsn_segsum_prepare_hip)
  * gsn_linear_fwd_hip       the same stage for shapes outside the fused kernel (any K / n_out, elu / tanh)
  * gsn_propagate_fwd_hip    the stand-alone scatter-add (and the gin / ogb message assembly)
  * gsn_code_stage_fwd_hip   first Linear over integer-coded inputs as a weight-row gather (``Codes``)
Restructuring that only changes fp32 rounding order (tolerance 1e-5, tests/test_layers_gpu.py): for
``msg_kind='general'`` the last Linear of ``msg_fn`` is applied after the sum aggregation,
``sum_e (W r_e + b) = W (sum_e r_e) + deg * b`` (SURVEY.md 7 "design notes for the MP kernels").
Backward (training) = kernels with their own adjoints composed under autograd: gsn_propagate_bwd_hip (scatter-add),
gsn_bn_act_bwd_hip + gsn_wgrad_hip + the forward kernel on W^T (dense stages), gsn_gather_cat_hip (edge rows of the
general layers).  A PyTorch re-computation remains only for gradients with BatchNorm in eval mode.
There is no CPU path: calling a layer on CPU tensors raises.
"""
from __future__ import annotations

import ctypes
import os
import time
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _abi, packs

__all__ = ["mlp", "central_encoder", "GSN_sparse", "GSN_edge_sparse", "MPNN_sparse", "MPNN_edge_sparse",
           "GSN_edge_sparse_ogb", "MPNN_edge_sparse_ogb", "build_csr", "propagate", "run_stages", "one_hot_identifiers",
           "global_add_pool_sparse", "global_mean_pool_sparse", "Codes", "run_linear_module", "invalidate_caches", "drop_input_caches", "drop_capture_caches", "set_graph_partition", "build_csr_graphs"]

_ACT_CODE = {"identity": 0, "relu": 1, "elu": 2, "tanh": 3}
_MAX_BLOCKS = 5

# Optional per-kernel timing hook for bench.py: a dict name -> list of (start_event, end_event, work) recorded on the
# launch stream around every kernel family; None = off (no events, no overhead).
KERNEL_TIMER = None
KERNEL_TIMER_ONLY = None       # a set of family names: only those are bracketed (an event pair per launch is not free: bench.py)


class _timed:
    def __init__(self, name, work=0.0):
        self.name, self.work = name, work
        self.on = KERNEL_TIMER is not None and (KERNEL_TIMER_ONLY is None or name in KERNEL_TIMER_ONLY)

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            KERNEL_TIMER.setdefault(self.name, []).append((self.e0, self.e1, self.work))
        return False


def choose_activation(activation):
    """models_misc.py:5-15"""
    if activation == "elu":
        return nn.ELU()
    if activation == "relu":
        return nn.ReLU()
    if activation == "tanh":
        return nn.Tanh()
    if activation == "identity":
        return lambda x: x
    raise NotImplementedError


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("gsn_amd.layers: %s is on %s; the layers run on HIP kernels only (no CPU fallback)" % (what, t.device))


# Zero-initialised scratch (fp64 column statistics, gradient accumulators of the atomically-adding kernels): handed out as slices of a
# 256 KiB arena that ONE fill zeroes, instead of a fill launch per request -- at the reference's batch sizes a training step asked for
# ~45 such buffers of 1-2 KiB, 5 us of launch each.  A slice is handed out once; the arena lives as long as any slice of it.  Keyed on
# (device, stream, capture id): the fill runs on the stream the consumers run on, and an arena filled inside one graph capture is
# never used by another capture or by eager launches (its fill is a node of that graph only).
_ZARENA = {}
ZERO_ARENA = os.environ.get("GSN_ZERO_ARENA", "1") != "0"      # (0: every request is its own torch.zeros -- A/B and fault isolation)
_ZARENA_TIERS = ((256 * 1024, 64 * 1024), (8 * 1024 * 1024, 2 * 1024 * 1024))     # (arena bytes, largest request served from it)


_ITEMSIZE = {torch.float64: 8, torch.float32: 4, torch.int64: 8, torch.int32: 4, torch.float16: 2, torch.uint8: 1}


def _zeros(n, dtype, device):
    """1-D zero tensor of ``n`` elements of ``dtype`` on ``device`` (cuda) from the arenas: small requests (statistics, status words) from a
    256 KiB arena, the weight-gradient accumulators of a dense backward (up to 2 MiB) from an 8 MiB one -- a d = 300 training step asks for
    ~20 of those, one fill of 8 MiB costs what one fill of 700 KiB does."""
    item = _ITEMSIZE[dtype]
    nbytes = (n * item + 255) // 256 * 256
    tier = 0 if nbytes <= 65536 else (1 if nbytes <= 2097152 else -1)
    if tier < 0 or device.type != "cuda" or not ZERO_ARENA:
        return torch.zeros(n, dtype=dtype, device=device)
    idx = device.index
    if idx is None:
        idx = torch._C._cuda_getDevice()
    stream = torch._C._cuda_getCurrentRawStream(idx)
    # (the capture id is asked of the library only while PyTorch says a capture is under way: this runs ~50 times per training step)
    if idx == torch._C._cuda_getDevice():
        cap = int(_abi.lib().gsn_stream_capture_id(stream)) if torch._C._cuda_isCurrentStreamCapturing() else 0
    else:       # (not the current device: PyTorch's query is about the current one)
        with _abi.device_guard(device):
            cap = int(_abi.lib().gsn_stream_capture_id(stream))
    hit = _ZARENA.get((idx, tier))
    if hit is None or hit[0] != stream or hit[3] != cap or hit[2] + nbytes > _ZARENA_TIERS[tier][0]:
        with _abi.device_guard(device):
            hit = [stream, torch.zeros(_ZARENA_TIERS[tier][0], dtype=torch.uint8, device=device), 0, cap]
        _ZARENA[(idx, tier)] = hit
    off = hit[2]
    hit[2] = off + nbytes
    return hit[1][off:off + n * item].view(dtype)


def _f32c(t):
    if t.dtype is torch.float32 and t.is_contiguous():
        return t.detach() if t.requires_grad else t
    return t.detach().to(torch.float32).contiguous()


class Codes:
    """Integer category codes standing in for their one-hot encoding (the output of the reference's
    DiscreteEmbedding('one_hot_encoder'), utils_graph_learning.py:170-187) as a layer input.

    ``codes`` int64 [R, C] on the GPU, ``n_classes`` C ints; equivalent to the float tensor ``dense()`` of shape
    [R, sum(n_classes)].  Layers accept it for ``x``, ``identifiers`` and ``edge_features``; where all inputs of msg_fn's
    first Linear are Codes that Linear becomes a weight-row gather (gsn_code_stage_fwd_hip) and the dense one-hot
    matrix is never built; everywhere else the layer densifies it."""
    __slots__ = ("codes", "n_classes", "clamp", "_dense", "_pack16", "__weakref__")

    def __init__(self, codes, n_classes, clamp=False):
        codes = codes.unsqueeze(-1) if codes.dim() == 1 else codes
        _need_cuda(codes, "codes")
        self.codes = codes.to(torch.int64).contiguous()
        self.n_classes = [int(c) for c in n_classes]
        if len(self.n_classes) != self.codes.shape[1]:
            raise ValueError("Codes: %d columns but %d class counts" % (self.codes.shape[1], len(self.n_classes)))
        self.clamp = bool(clamp)      # values above the last class count as the last class (as gsn_one_hot_hip's clamp)
        self._pack16 = None           # (pack, first column) once gsn_amd.packs has encoded these codes into an exact fp16 row pack
        self._dense = None

    @property
    def shape(self):
        return torch.Size((self.codes.shape[0], sum(self.n_classes)))

    @property
    def device(self):
        return self.codes.device

    is_cuda = True
    requires_grad = False

    def dim(self):
        return 2

    def dense(self):
        # (kept with the code tensor's version counter: a reused input buffer rewritten in place is encoded again)
        if self._dense is None or self._dense[1] != self.codes._version:
            self._dense = (one_hot_identifiers(self.codes, self.n_classes, clamp=self.clamp), self.codes._version)
        return self._dense[0]


def _dense(v):
    return v.dense() if isinstance(v, Codes) else v


# ------------------------------------------------------------------------------------------------------------------
# CSR of the aggregation index (cached per edge_index tensor)
# ------------------------------------------------------------------------------------------------------------------
class _CSR:
    """Target-sorted CSR of one aggregation index.  ``deg`` / ``deg4`` (in-degree as a float column, and the same padded to
    four columns) are only needed by the multi-launch path of the `general` layers and are built on first use: the one-launch
    layer kernel takes the degrees from ``seg_ptr`` itself."""
    __slots__ = ("seg_ptr", "perm", "tgt", "src", "_deg", "_deg4", "part")

    @property
    def deg(self):
        if self._deg is None:
            self._deg = (self.seg_ptr[1:] - self.seg_ptr[:-1]).to(torch.float32).unsqueeze(1).contiguous()
        return self._deg

    @property
    def deg4(self):
        if self._deg4 is None:
            self._deg4 = torch.nn.functional.pad(self.deg, (0, 3))
        return self._deg4


_CSR_CACHE = {}


def build_csr(index, n_nodes, with_targets=False, other=None):
    """(seg_ptr int32 [N+1], perm int32 [E]) grouping edge ids by ``index`` (stable), via gsn_csr_build_hip;
    with_targets: also sorted_target int32 [E] = index[perm] (and sorted_other = other[perm] if ``other`` is given)."""
    _need_cuda(index, "edge_index")
    index = index.contiguous()
    E = index.numel()
    L = _abi.lib()
    dev = index.device
    seg_ptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    scratch = torch.empty(int(L.gsn_csr_scratch_elems(n_nodes)), dtype=torch.int32, device=dev)
    tgt = torch.empty(max(E, 1), dtype=torch.int32, device=dev) if with_targets else None
    src = torch.empty(max(E, 1), dtype=torch.int32, device=dev) if (with_targets and other is not None) else None
    if other is not None:
        other = other.contiguous()
    with _abi.device_guard(dev), _timed("csr_build", 12.0 * E + 8.0 * n_nodes):
        _abi.check(L.gsn_csr_build_hip(n_nodes, E, index.data_ptr() if E else None,
                                       other.data_ptr() if (other is not None and E) else None, seg_ptr.data_ptr(),
                                       perm.data_ptr(), _abi.ptr(tgt), _abi.ptr(src) if (E and other is not None) else None,
                                       scratch.data_ptr(), _abi.current_stream()), "gsn_csr_build_hip")
    if with_targets:
        return seg_ptr, perm[:E], tgt[:E], (src[:E] if src is not None else None)
    return seg_ptr, perm[:E]


def build_csr_graphs(index, n_nodes, node_ptr, edge_ptr, max_nodes, max_edges, other=None, check=True):
    """:func:`build_csr` (with targets) for a collated batch whose graph boundaries are known: ONE launch, every graph sorted in
    LDS (gsn_csr_build_graphs_hip).  ``node_ptr`` / ``edge_ptr``: int64 device [G + 1].  ``check``: read the status word back
    (a column that leaves its graph's vertex range means the pointers do not describe this batch -> ValueError)."""
    _need_cuda(index, "edge_index")
    index = index.contiguous()
    E = index.numel()
    dev = index.device
    seg_ptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    tgt = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    src = torch.empty(max(E, 1), dtype=torch.int32, device=dev) if other is not None else None
    status = _zeros(1, torch.int32, dev)
    if other is not None:
        other = other.contiguous()
    G = node_ptr.numel() - 1
    with _abi.device_guard(dev), _timed("csr_build", 28.0 * E + 4.0 * n_nodes):
        _abi.check(_abi.lib().gsn_csr_build_graphs_hip(G, node_ptr.data_ptr(), edge_ptr.data_ptr(), n_nodes, E, int(max_nodes), int(max_edges),
                                                       index.data_ptr() if E else None, other.data_ptr() if (other is not None and E) else None,
                                                       seg_ptr.data_ptr(), perm.data_ptr(), tgt.data_ptr(), _abi.ptr(src) if E else None,
                                                       status.data_ptr(), _abi.current_stream()), "gsn_csr_build_graphs_hip")
    if check:
        st = int(status.item())
        if st:
            raise ValueError("build_csr_graphs: node_ptr / edge_ptr do not describe this edge_index (status %d)" % st)
    return seg_ptr, perm[:E], tgt[:E], (src[:E] if src is not None else None)


_PARTITION = {}
_CSR_GRAPHS_LDS = 64 * 1024


def set_graph_partition(edge_index, node_ptr, edge_ptr, max_nodes, max_edges, check=True):
    """Tell the layers that ``edge_index`` (the tensor object later passed to ``forward``) is a collated batch with these graph
    boundaries (int64 device [G + 1]; what torch_geometric's ``Batch.ptr`` and the counting kernel's pointers hold): its
    aggregation index is then built by one launch per batch instead of the generic seven (the reference has no counterpart:
    it re-sorts a COO tensor in every layer, GSN_sparse.py:140-143).  Without this call nothing changes."""
    _need_cuda(edge_index, "edge_index")
    if (2 * (int(max_nodes) + 1) + 2 * int(max_edges)) * 4 > _CSR_GRAPHS_LDS:
        return False                       # graphs too large for the per-graph kernel: the generic build is used
    key = id(edge_index)

    def _gone(_ref, key=key):
        cache = _PARTITION
        if cache is None:                   # (interpreter shutdown: module globals are already cleared)
            return
        hit = cache.get(key)
        if hit is not None and hit[0] is _ref:
            del cache[key]
    _PARTITION[key] = (weakref.ref(edge_index, _gone), edge_index._version,
                       (node_ptr.to(device=edge_index.device, dtype=torch.int64).contiguous(),
                        edge_ptr.to(device=edge_index.device, dtype=torch.int64).contiguous(), int(max_nodes), int(max_edges), bool(check)))
    return True


def _partition_of(edge_index):
    hit = _PARTITION.get(id(edge_index))
    if hit is not None and hit[0]() is edge_index and hit[1] == edge_index._version:
        return hit[2]
    return None


_BATCH_PTR = {}


def set_batch_partition(batch, node_ptr):
    """Tell the readout that ``batch`` (the tensor object later passed to the pooling functions) is the SORTED graph-id vector of
    a collated batch with these boundaries (int64 [G + 1]: ``Batch.ptr``): its rows are already grouped by graph, so the
    segmented sum needs no index build at all (the generic build sorts the N row ids by graph id with seven launches)."""
    _need_cuda(batch, "batch")
    key = id(batch)

    def _gone(_ref, key=key):
        cache = _BATCH_PTR
        if cache is None:                   # (interpreter shutdown: module globals are already cleared)
            return
        hit = cache.get(key)
        if hit is not None and hit[0] is _ref:
            del cache[key]
    _BATCH_PTR[key] = (weakref.ref(batch, _gone), batch._version, node_ptr.to(device=batch.device, dtype=torch.int32).contiguous())


def _batch_ptr_of(batch):
    hit = _BATCH_PTR.get(id(batch))
    if hit is not None and hit[0]() is batch and hit[1] == batch._version:
        return hit[2]
    return None


def _cache_put(key, owner, value):
    """_CSR_CACHE entry that disappears with the tensor it belongs to (weak-reference callback), so batches that are
    dropped do not leave E-sized index tensors behind."""
    def _gone(_ref, key=key):
        cache = _CSR_CACHE
        if cache is None:                   # (interpreter shutdown: module globals are already cleared)
            return
        hit = cache.get(key)
        if hit is not None and hit[0] is _ref:
            del cache[key]
    ref = weakref.ref(owner, _gone)
    _CSR_CACHE[key] = (ref, owner._version, value)


def _cache_get(key, owner):
    hit = _CSR_CACHE.get(key)
    if hit is not None:
        ref, version, value = hit
        if ref() is owner and version == owner._version:
            return value
    return None


def _csr_for(edge_index, row, n_nodes):
    """CSR of ``edge_index[row]`` cached on the tensor OBJECT (weak reference + version counter): a freed tensor's address
    is reused by the caching allocator, so (data_ptr, shape) alone would return a stale CSR for a different graph of the
    same size (e.g. the 15 SR(25,12,5,6) graphs all have E = 300)."""
    key = (id(edge_index), row, n_nodes)
    c = _cache_get(key, edge_index)
    if c is not None:
        return c
    c = _CSR()
    part = _partition_of(edge_index)
    c.part = part                           # (graph boundaries of a collated batch: the graph-aligned d = 128 layer kernel reads them)
    if part is not None:
        c.seg_ptr, c.perm, c.tgt, c.src = build_csr_graphs(edge_index[row], n_nodes, part[0], part[1], part[2], part[3],
                                                           other=edge_index[1 - row], check=part[4])
    else:
        c.seg_ptr, c.perm, c.tgt, c.src = build_csr(edge_index[row], n_nodes, with_targets=True, other=edge_index[1 - row])
    c._deg = c._deg4 = None
    _cache_put(key, edge_index, c)
    return c


def num_graphs_of(batch):
    """1 + the largest graph id of a ``batch`` vector -- ONE device read per batch tensor (cached on the tensor object with its version
    counter; a registered partition answers without any): the reference reads it once per readout, a host synchronisation each time."""
    ptr = _batch_ptr_of(batch)
    if ptr is not None:
        return int(ptr.numel()) - 1
    key = (id(batch), "n_graphs")
    g = _cache_get(key, batch)
    if g is None:
        g = int(batch.max().item()) + 1 if batch.numel() else 0
        _cache_put(key, batch, g)
    return g


def global_add_pool_sparse(x, batch, num_graphs=None):
    """Sum readout (utils_graph_learning.py:23-29: COO [G, N, d] + torch.sparse.sum) as a segmented sum keyed by the
    ``batch`` vector, on the propagate kernel.  The (row id, graph id) index pair is cached on the
    ``batch`` tensor, so repeated readouts of one batch (every layer of a jumping-knowledge model) build its CSR once."""
    _need_cuda(x, "x")
    n_rows = x.shape[0]
    if batch.numel() != n_rows:
        raise RuntimeError("global_add_pool_sparse: %d rows but %d batch entries" % (n_rows, batch.numel()))
    g = num_graphs_of(batch) if num_graphs is None else int(num_graphs)
    key = (id(batch), "pool", n_rows)
    ei = _cache_get(key, batch)
    if ei is None:
        # rows are "edges" whose target is their graph id; the message is the row itself
        ei = torch.stack([torch.arange(n_rows, device=x.device, dtype=torch.int64), batch.to(torch.int64)], 0)
        _cache_put(key, batch, ei)
    ptr = _batch_ptr_of(batch)
    if ptr is not None and ptr.numel() == g + 1 and _cache_get((id(ei), 1, g), ei) is None:
        # rows grouped by graph already: segment g = rows ptr[g] .. ptr[g + 1], in place
        c = _CSR()
        c.seg_ptr = ptr
        c.perm = torch.arange(n_rows, device=x.device, dtype=torch.int32)
        c.tgt, c.src = batch.to(torch.int32), c.perm
        c._deg = c._deg4 = None
        _cache_put((id(ei), 1, g), ei, c)
    return propagate(0, ei, 1, g, b=x)


class _AddByGraphFn(torch.autograd.Function):
    """x + table[batch] in one pass (gsn_add_gathered_hip); adjoint: identity for x, the sum readout for the table."""

    @staticmethod
    def forward(ctx, x, table, batch):
        xs, ts = _f32c(x), _f32c(table)
        idx = batch.to(torch.int64).contiguous()
        out = torch.empty_like(xs)
        with _abi.device_guard(xs.device), _timed("add_gathered", 12.0 * xs.numel()):
            rc = _abi.lib().gsn_add_gathered_hip(xs.shape[0], xs.shape[1], xs.data_ptr() if xs.numel() else None, _abi.ptr(ts),
                                                 idx.data_ptr() if idx.numel() else None, ts.shape[0], out.data_ptr() if out.numel() else None,
                                                 _abi.current_stream())
        _abi.check(rc, "gsn_add_gathered_hip")
        ctx.batch, ctx.n_table = batch, ts.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        g_table = None
        if ctx.needs_input_grad[1]:
            with torch.no_grad():
                g_table = global_add_pool_sparse(g, ctx.batch, ctx.n_table)
        return (g if ctx.needs_input_grad[0] else None), g_table, None


def add_by_graph(x, table, batch):
    """``x + table[batch]`` (models_graph_classification_ogb_original.py:236: the virtual node's embedding joins every vertex of its
    graph) as one kernel, with the readout kernel as the adjoint of the gather."""
    _need_cuda(x, "x")
    if x.dim() != 2 or table.dim() != 2 or x.shape[1] != table.shape[1] or batch.numel() != x.shape[0]:
        return x + table[batch]            # (shapes the reference would broadcast or reject: its own expression)
    return _AddByGraphFn.apply(x, table, batch)


def global_mean_pool_sparse(x, batch, num_graphs=None):
    """Mean readout (utils_graph_learning.py:32-41): sum readout divided by the graph sizes (empty graphs divide by 1)."""
    s = global_add_pool_sparse(x, batch, num_graphs)
    # max(size, 1) per graph: a property of the batch vector, kept with it (torch.bincount sizes its output from a device read --
    # a host synchronisation per readout, and not capturable: gsn_amd.graphs)
    key = (id(batch), "sizes", s.shape[0], s.dtype)
    inv = _cache_get(key, batch)
    if inv is None:
        sizes = torch.zeros(s.shape[0], dtype=s.dtype, device=s.device)
        if batch.numel():
            sizes.index_add_(0, batch.to(torch.int64), torch.ones(batch.numel(), dtype=s.dtype, device=s.device))
        inv = sizes.clamp_(min=1.0).unsqueeze(1)
        _cache_put(key, batch, inv)
    return s / inv


def one_hot_identifiers(values, n_classes, clamp=False):
    """Multi-hot float encoding of integer identifier columns on the device (gsn_one_hot_hip): the reference's
    one_hot_encoder (utils_graph_learning.py:170-187).  values: int64 [M, C] cuda; n_classes: list of C ints."""
    import numpy as np
    _need_cuda(values, "identifiers")
    values = values.to(torch.int64).contiguous()
    if values.dim() == 1:
        values = values.unsqueeze(-1)
    ncls = np.ascontiguousarray(n_classes, dtype=np.int32)
    if len(ncls) != values.shape[1]:
        raise ValueError("one_hot_identifiers: %d columns but %d class counts" % (values.shape[1], len(ncls)))
    out = torch.empty((values.shape[0], int(ncls.sum())), dtype=torch.float32, device=values.device)
    with _abi.device_guard(values.device), _timed("one_hot", 8.0 * values.numel() + 4.0 * out.numel()):
        _abi.check(_abi.lib().gsn_one_hot_hip(values.shape[0], values.shape[1], values.data_ptr(), _abi.ptr(ncls), int(bool(clamp)),
                                              out.data_ptr(), _abi.current_stream()), "gsn_one_hot_hip")
    return out


# ------------------------------------------------------------------------------------------------------------------
# propagate (scatter-add with fused message assembly) -- HIP forward and HIP adjoint
# ------------------------------------------------------------------------------------------------------------------
class _PropagateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, edge_index, sel, n_nodes, b_per_node, a, b, c, pads, eps, *selfs):
        # selfs: the layer's own term -- blocks [N][w] or [1][w] (one row for every vertex); pads: zero columns in front of b / c
        tgt_row, src_row = sel, 1 - sel
        csr_t = _csr_for(edge_index, tgt_row, n_nodes)
        src = edge_index[src_row].contiguous()
        E = src.numel()
        ts = [None if t is None else _f32c(t) for t in (a, b, c)]
        widths = [0 if t is None else t.shape[1] for t in ts]
        pad_b, pad_c = (pads[0] if widths[1] else 0), (pads[1] if widths[2] else 0)
        ss = [_f32c(t) for t in selfs]
        d_out = (sum(widths) + pad_b + pad_c) if kind == 0 else max(widths)
        if ss and kind == 0 and sum(t.shape[1] for t in ss) != d_out:
            raise RuntimeError("propagate: the self blocks are %d columns wide, the messages %d" % (sum(t.shape[1] for t in ss), d_out))
        out = torch.empty((n_nodes, d_out), dtype=torch.float32, device=edge_index.device)
        # algorithmic bytes: src (8) + perm (4) per edge, every message element read once, output written once
        per_edge = (0 if ts[0] is None else widths[0]) + (0 if (ts[1] is None or b_per_node) else widths[1]) + (0 if ts[2] is None else widths[2])
        bytes_alg = 12.0 * E + 4.0 * n_nodes + 4.0 * (E * per_edge + n_nodes * d_out) + 4.0 * sum(t.numel() for t in ss)
        arr = (_abi.gsn_self_block * max(1, len(ss)))()
        for k, t in enumerate(ss):
            if t.shape[0] not in (1, n_nodes):
                raise RuntimeError("propagate: self block %d has %d rows (1 or %d expected)" % (k, t.shape[0], n_nodes))
            arr[k].data = t.data_ptr(); arr[k].width = t.shape[1]; arr[k].row_stride = 0 if (t.shape[0] == 1 and n_nodes != 1) else t.shape[1]
        eps32 = None if eps is None else _f32c(eps.reshape(-1))
        with _abi.device_guard(edge_index.device), _timed("propagate_fwd", bytes_alg):
            rc = _abi.lib().gsn_propagate_self_fwd_hip(kind, n_nodes, E, src.data_ptr() if E else None, csr_t.seg_ptr.data_ptr(),
                                                       csr_t.perm.data_ptr() if E else None,
                                                       csr_t.src.data_ptr() if (E and csr_t.src is not None) else None,
                                                       _abi.ptr(ts[0]), widths[0], _abi.ptr(ts[1]), widths[1], int(b_per_node),
                                                       _abi.ptr(ts[2]), widths[2], pad_b, pad_c, len(ss), arr, _abi.ptr(eps32),
                                                       out.data_ptr(), _abi.current_stream())
        _abi.check(rc, "gsn_propagate_self_fwd_hip")
        ctx.kind, ctx.sel, ctx.n_nodes, ctx.b_per_node = kind, sel, n_nodes, b_per_node
        ctx.edge_index = edge_index
        ctx.widths, ctx.pads, ctx.n_self, ctx.has_eps = widths, (pad_b, pad_c), len(ss), eps is not None
        ctx.eps_shape = None if eps is None else eps.shape
        empty = torch.empty(0, device=edge_index.device)
        ctx.save_for_backward(*[t if t is not None else empty for t in ts], eps32 if eps is not None else empty, *ss)
        return out

    @staticmethod
    def backward(ctx, g_out):
        saved = ctx.saved_tensors
        a, b, c = [t if t.numel() else None for t in saved[:3]]
        eps32, ss = saved[3], saved[4:]
        ei, sel, n = ctx.edge_index, ctx.sel, ctx.n_nodes
        src = ei[1 - sel].contiguous()
        tgt = ei[sel].contiguous()
        E = src.numel()
        g_out = _f32c(g_out)
        need = ctx.needs_input_grad[5:8]
        dev = ei.device
        wa, wb, wc = ctx.widths
        # the source-sorted CSR is only needed for per-node gradients (g_a, per-node g_b)
        need_node = (need[0] and wa) or (need[1] and wb and ctx.b_per_node)
        csr_s = _csr_for(ei, 1 - sel, n) if need_node else None
        # (every element of the three is written by the kernels: no zero fill)
        g_a = torch.empty((n, wa), dtype=torch.float32, device=dev) if (need[0] and wa) else None
        g_b = None
        if need[1] and wb:
            g_b = torch.empty((n if ctx.b_per_node else E, wb), dtype=torch.float32, device=dev)
        g_c = torch.empty((E, wc), dtype=torch.float32, device=dev) if (need[2] and wc) else None
        # relu-sum: the per-edge gradients of b and c are the SAME rows (relu'(a_j + b + c) g_out[t]) -- written once, handed to both
        shared_bc = ctx.kind == 1 and g_c is not None and g_b is not None and not ctx.b_per_node
        if shared_bc:
            g_b = None
        if g_a is not None or g_b is not None or g_c is not None:
            with _abi.device_guard(dev):
                rc = _abi.lib().gsn_propagate_pad_bwd_hip(ctx.kind, n, E, src.data_ptr() if E else None, tgt.data_ptr() if E else None,
                                                          csr_s.seg_ptr.data_ptr() if csr_s is not None else None,
                                                          csr_s.perm.data_ptr() if (csr_s is not None and E) else None,
                                                          _abi.ptr(a), wa, _abi.ptr(b), wb, int(ctx.b_per_node), _abi.ptr(c), wc,
                                                          ctx.pads[0], ctx.pads[1], g_out.data_ptr(), _abi.ptr(g_a), _abi.ptr(g_b),
                                                          _abi.ptr(g_c), _abi.current_stream())
            _abi.check(rc, "gsn_propagate_pad_bwd_hip")
        if shared_bc:
            g_b = g_c
        # the self term (1 + eps) * self: one pass over g_out (gsn_propagate_self_bwd_hip)
        g_eps, g_selfs = None, [None] * ctx.n_self
        want_eps = ctx.has_eps and ctx.needs_input_grad[9]
        want_self = [bool(ctx.needs_input_grad[10 + k]) for k in range(ctx.n_self)]
        if ctx.n_self and (want_eps or any(want_self)):
            d_out = g_out.shape[1]
            single = [t.shape[0] == 1 and n != 1 for t in ss]
            arr = (_abi.gsn_self_block * ctx.n_self)()
            gptr = (_abi.c_vp * ctx.n_self)()
            for k, t in enumerate(ss):
                arr[k].data = t.data_ptr(); arr[k].width = t.shape[1]; arr[k].row_stride = 0 if single[k] else t.shape[1]
                if want_self[k] and not single[k]:
                    g_selfs[k] = torch.empty((n, t.shape[1]), dtype=torch.float32, device=dev)
                gptr[k] = None if g_selfs[k] is None else g_selfs[k].data_ptr()
            need_col = any(w and sg for w, sg in zip(want_self, single))
            acc = _zeros(1 + (d_out if need_col else 0), torch.float64, dev)
            with _abi.device_guard(dev):
                rc = _abi.lib().gsn_propagate_self_bwd_hip(ctx.kind, n, d_out, g_out.data_ptr(), ctx.n_self, arr, gptr,
                                                           eps32.data_ptr() if ctx.has_eps else None, acc.data_ptr() if want_eps else None,
                                                           acc.data_ptr() + 8 if need_col else None, _abi.current_stream())
            _abi.check(rc, "gsn_propagate_self_bwd_hip")
            if want_eps:
                g_eps = acc[0].to(torch.float32).reshape(ctx.eps_shape)
            o = 0
            for k, t in enumerate(ss):
                w = t.shape[1]
                if want_self[k] and single[k]:
                    g_selfs[k] = (acc[1 + o:1 + o + w] if ctx.kind == 0 else acc[1:1 + d_out]).to(torch.float32).reshape(1, w)
                if ctx.kind == 0:
                    o += w
        return (None, None, None, None, None, g_a, g_b, g_c, None, g_eps) + tuple(g_selfs)


def propagate(kind, edge_index, sel, n_nodes, a=None, b=None, c=None, b_per_node=False, selfs=(), eps=None, pads=(0, 0)):
    """out[t] = [(1 + eps) * self[t] +] sum_{e: edge_index[sel, e] = t} msg_e  with msg_e = cat(a[src_e], 0.., b, 0.., c) (kind 0; ``pads``
    zero columns in front of b and c) or relu(a[src_e] + b + c) (kind 1); b is per edge, or per node gathered at src if ``b_per_node``.
    ``selfs``: blocks of the layer's own term, concatenated (kind 0) or added (kind 1), each [N][w] or [1][w] (the same row for every
    vertex); ``eps`` a 0-dim / 1-element tensor (GSN_sparse.py:157-163, GSN_edge_sparse_ogb.py:103-106)."""
    return _PropagateFn.apply(kind, edge_index, sel, n_nodes, bool(b_per_node), a, b, c, (int(pads[0]), int(pads[1])), eps, *selfs)


# ------------------------------------------------------------------------------------------------------------------
# fused Linear (+BN) (+activation) stage
# ------------------------------------------------------------------------------------------------------------------
LINEAR_F16X3 = os.environ.get("GSN_LINEAR_F16X3", "1") != "0"     # direct-row dense stages on the fp16x3 kernel (else bf16x6 / fp32)
LINEAR_F16X3_MIN_N = int(os.environ.get("GSN_LINEAR_F16X3_MIN_N", "128"))
# products of at most this many 128 x 128 output tiles stay on the bf16x6 kernel (its 32-row-tile twin, csrc/linear.hip): one launch of ~10-19 us
# instead of weight split + row pre-pass + product = three launches of ~20-30 us together (molhiv B = 32: 837 x 300 -> 600)
LINEAR_F16X3_MIN_TILES = int(os.environ.get("GSN_LINEAR_F16X3_MIN_TILES", "96"))
# train-mode BatchNorm stages too: the pre-BN rows AND their fp64 column statistics from one launch of the fp16x3 kernel
# (gsn_linear_f16x3_fwd_stats_hip: linear_fwd_bf16_kernel<STATS>'s contract at half its matrix work -- 105 k x 300 -> 600: 0.27 -> 0.18 ms with
# the row pre-pass; rows as accurate as the bf16x6 kernel's against fp64, scripts/gpu/stats_ab.py).  0: those stages stay on the bf16x6 kernel
LINEAR_F16X3_STATS = os.environ.get("GSN_LINEAR_F16X3_STATS", "1") != "0"


STRIDED_WEIGHTS = os.environ.get("GSN_STRIDED_WEIGHTS", "1") != "0"      # transposed weight views read through their strides (0: a contiguous copy first)


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


def _linear_hip(blocks, weight, bias, bn_mean, bn_scale, bn_shift, act, m_rows, out=True, stats=None):
    """blocks: list of (data [R,w] fp32 cuda, idx int64 [M] or None)."""
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
              and weight.stride(1) >= weight.shape[0] and STRIDED_WEIGHTS)
    w = (weight.detach() if weight.requires_grad else weight) if w_view else _f32c(weight)
    vecs = [None if v is None else _f32c(v) for v in (bias, bn_mean, bn_scale, bn_shift)]
    # direct rows (node-level stages): the fp16x3 kernel with the weights split once per weight version
    # (from two column tiles on: the pre-pass over the rows that finds their scales is then amortised -- at n_out <= 128 the
    #  bf16x6 kernel, which reads the rows once, is faster: 99 vs 90 TF/s at K = 260)
    # (a train-mode stage that keeps its pre-BN rows: the same kernel with the column statistics taken in its epilogue)
    if (LINEAR_F16X3 and out and (stats is None or (LINEAR_F16X3_STATS and bn_mean is None and act == 0 and n_out % 4 == 0)) and m_rows > 0 and n_out > LINEAR_F16X3_MIN_N
            and ((m_rows + 127) // 128) * ((n_out + 127) // 128) > LINEAR_F16X3_MIN_TILES
            and all(idx is None for _, idx in blocks) and all(d.shape[1] % 4 == 0 and d.data_ptr() % 16 == 0 for d in keep)):
        planes, col_inv = _f16x3_weights(weight, w)
        scratch = torch.empty(int(_abi.lib().gsn_linear_f16x3_scratch_bytes(m_rows, w.shape[1])), dtype=torch.uint8, device=dev)
        with _abi.device_guard(dev), _timed("linear_fwd", 2.0 * m_rows * w.shape[1] * n_out):
            if stats is not None:
                rc = _abi.lib().gsn_linear_f16x3_fwd_stats_hip(m_rows, len(blocks), arr, planes.data_ptr(), col_inv.data_ptr(), _abi.ptr(vecs[0]), n_out,
                                                               scratch.data_ptr(), y.data_ptr(), stats.data_ptr(), _abi.current_stream())
            else:
                rc = _abi.lib().gsn_linear_f16x3_fwd_hip(m_rows, len(blocks), arr, planes.data_ptr(), col_inv.data_ptr(), _abi.ptr(vecs[0]), n_out,
                                                         _abi.ptr(vecs[1]), _abi.ptr(vecs[2]), _abi.ptr(vecs[3]), act, scratch.data_ptr(),
                                                         y.data_ptr(), _abi.current_stream())
        _abi.check(rc, "gsn_linear_f16x3_fwd_stats_hip" if stats is not None else "gsn_linear_f16x3_fwd_hip")
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


VALIDATE_CACHES = os.environ.get("GSN_VALIDATE_CACHES", "0") != "0"   # re-derive-and-compare mode for the per-weight caches (below)


def _module_fingerprint(module):
    """Validation mode (``GSN_VALIDATE_CACHES=1`` / ``layers.VALIDATE_CACHES = True``): a content fingerprint of every floating-point
    parameter and buffer of ``module`` -- three moments per tensor, ONE read-back per forward.  The derived-weight caches (prepared
    fp16 fragments of the one-launch layer, folded first weight, fp16 planes of the dense stages, eval-mode BatchNorm vectors) are
    keyed on ``tensor._version`` and ``data_ptr``, which a write through ``.data`` (EMA / SWA ``p.data.copy_``, weight clipping,
    manual surgery) does not change; with this mode on, such a write is noticed at the next forward and the caches of the module are
    dropped.  Costs a device synchronisation per layer forward: a debugging / validation switch, off by default -- production code
    that writes through ``.data`` calls :func:`invalidate_caches` instead."""
    vals = []
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_floating_point() and t.numel():
            f = t.detach().reshape(-1).double()
            w = torch.arange(1, f.numel() + 1, device=f.device, dtype=torch.float64)
            vals += [f.sum(), (f * f).sum(), (f * w).sum()]
    return tuple(torch.stack(vals).tolist()) if vals else ()


# Asynchronous validation of the per-weight caches (default on; GSN_ASYNC_VALIDATE=0 turns it off).  Behind every eval-mode forward of a
# layer ONE kernel fingerprints the layer's parameters and buffers (gsn_fingerprint_hip); the 8 bytes travel to pinned host memory behind
# it and are looked at -- without waiting -- at the layer's next forward.  A fingerprint that moved while no version counter did is a
# write through `.data`: the layer's caches are dropped there and then and a RuntimeWarning names the layer.  The forward(s) between the
# write and that point used the old derived weights (the check costs no synchronisation; GSN_VALIDATE_CACHES=1 checks BEFORE every
# forward at the price of one); `invalidate_caches` after such a write remains the contract for code that cannot afford one stale call.
ASYNC_VALIDATE = os.environ.get("GSN_ASYNC_VALIDATE", "1") != "0"
# at most one fingerprint per layer and interval (seconds of wall clock): a tight inference loop pays one 5 us launch per layer every 20 ms, not
# one per forward; a `.data` write is then noticed within the interval plus one forward.  0: behind every eval forward.
ASYNC_VALIDATE_INTERVAL = float(os.environ.get("GSN_ASYNC_VALIDATE_INTERVAL", "0.02"))
RAW_WRITTEN = None      # a list while gsn_amd.graphs.GraphedTrainStep captures: tensors that captured kernels write through raw pointers
_FP_RING = [None, 0]


def _fp_slot():
    if _FP_RING[0] is None:
        _FP_RING[0] = torch.zeros(1024, dtype=torch.int64).pin_memory()
    i = _FP_RING[1]
    _FP_RING[1] = (i + 1) % 1024
    return _FP_RING[0][i:i + 1]


def _async_validate(module):
    st = module.__dict__.get("_gsn_fp_state")
    now = time.monotonic()
    if st is not None:
        pend = st["pending"]
        # (the common case of a tight loop: nothing landed, nothing due -- two dictionary reads and a clock)
        if now - st["t_last"] < ASYNC_VALIDATE_INTERVAL and (not pend or not pend[0][0].query()):
            return
        while pend and (len(pend) > 32 or pend[0][0].query()):
            ev, slot, vers = pend.pop(0)
            ev.synchronize()
            val = int(slot[0])
            last = st["last"]
            if last is not None and last[0] != val and last[1] == vers:
                import warnings
                invalidate_caches(module)
                warnings.warn("gsn_amd: a parameter or buffer of %s was written through `.data` (its version counter did not move): the forward(s) since "
                              "that write used weights prepared before it; the caches are dropped now (call gsn_amd.layers.invalidate_caches "
                              "after such a write, or set GSN_VALIDATE_CACHES=1)" % type(module).__name__, RuntimeWarning, stacklevel=3)
            st["last"] = (val, vers)
        if now - st["t_last"] < ASYNC_VALIDATE_INTERVAL:
            return
    tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.is_floating_point() and t.numel() and t.is_cuda and t.element_size() == 4]
    if not tensors:
        return
    dev = tensors[0].device
    ptrs = tuple(t.data_ptr() for t in tensors)
    versions = tuple(t._version for t in tensors)
    if st is None or st["ptrs"] != ptrs:
        meta = torch.tensor(list(ptrs) + [t.numel() for t in tensors], dtype=torch.int64).to(dev)       # (once per layer: parameters keep their addresses)
        st = {"ptrs": ptrs, "meta": meta, "max_words": max(t.numel() for t in tensors), "last": None, "pending": [], "t_last": -1e9}
        module.__dict__["_gsn_fp_state"] = st
    st["t_last"] = now
    acc = _zeros(1, torch.int64, dev)
    with _abi.device_guard(dev):
        slot = _fp_slot()
        _abi.check(_abi.lib().gsn_fingerprint_hip(len(tensors), st["meta"].data_ptr(), int(st["max_words"]), acc.data_ptr(), slot.data_ptr(),
                                                  _abi.current_stream()), "gsn_fingerprint_hip")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
    st["pending"].append((ev, slot, versions))


_CAPTURE_CACHED = []     # (owner, attribute) of every derived-weight cache entry made while a stream capture was under way


def _note_cache(owner, attr):
    """A derived tensor (prepared weights, folded weight, eval-mode BatchNorm vectors, transposed weight, fp16 planes) was just cached
    on ``owner``.  Made during a stream capture it lives in graph-pool memory that nothing has written until the first replay: noted, so
    that gsn_amd.graphs drops it behind the capture and an eager call before the first replay prepares its own."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        _CAPTURE_CACHED.append((weakref.ref(owner), attr))


def drop_capture_caches():
    """Drop the cache entries noted by _note_cache (called by gsn_amd.graphs right behind a capture)."""
    while _CAPTURE_CACHED:
        ref, attr = _CAPTURE_CACHED.pop()
        owner = ref()
        if owner is not None and hasattr(owner, attr):
            try:
                delattr(owner, attr)
            except AttributeError:
                pass


def drop_input_caches():
    """Drop everything cached per INPUT tensor (aggregation index of an ``edge_index``, readout index pairs and graph sizes of a
    ``batch`` vector): the next forward builds them again.  gsn_amd.graphs calls this in front of a stream capture."""
    # (the NUMBER of graphs of a batch vector is a shape, not contents: a captured step is bound to it anyway, and reading it again would
    #  synchronise inside the capture)
    for k in [k for k in _CSR_CACHE if not (isinstance(k, tuple) and len(k) == 2 and k[1] == "n_graphs")]:
        _CSR_CACHE.pop(k, None)              # (dropping an entry can free a tensor whose weak-reference callback removes another key)


def invalidate_caches(module=None):
    """Drop the derived tensors this module keeps per parameter VERSION (folded first weight of the `general` layers,
    eval-mode BatchNorm scale / shift vectors, transposed weights) -- needed only after writing a parameter or buffer
    through ``.data`` (``p.data.copy_`` / ``fill_``), which PyTorch does not count as a new version; optimizers,
    ``load_state_dict`` and ordinary in-place ops do bump the version and need no call.  ``module=None``: also the CSR cache."""
    if module is None:
        _CSR_CACHE.clear()
        return
    for m in module.modules():
        for attr in ("_fold_cache", "_split_cache", "_fused_prep", "_fused_prep16", "_gsn_eval_cache", "_gsn_wt"):
            if hasattr(m, attr):
                try:
                    delattr(m, attr)
                except AttributeError:
                    pass
        for prm in m.parameters(recurse=False):          # fp16 planes of the weights (gsn_linear_f16x3_prepare_hip)
            if hasattr(prm, "_gsn_f16x3"):
                del prm._gsn_f16x3


SPLIT_EDGE_STAGE = os.environ.get("GSN_SPLIT_EDGE", "1") != "0"        # node part of a wide edge Linear once per node (K > SPLIT_EDGE_MIN_K)
SPLIT_EDGE_MIN_K = int(os.environ.get("GSN_SPLIT_EDGE_MIN_K", "160"))
FUSED_LAYER = os.environ.get("GSN_LAYER_FUSED", "1") != "0"   # one-launch `general` layer (gsn_layer_fused_fwd_hip) where it fits


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


CHAIN_ROW_EXPONENTS = os.environ.get("GSN_CHAIN_ROW_EXP", "1") != "0"      # 128-wide one-launch layers leave their output's row exponents for the next layer


GRAPH_ALIGNED_LAYER = os.environ.get("GSN_LAYER_GRAPHS", "1") != "0"   # d = 128 layers of a collated batch: node products on graph-aligned tiles (csrc/layer_g.hip)
PACK16_LAYER = os.environ.get("GSN_LAYER_PACK16", "1") != "0"   # tagged exact inputs: the packed-row kernel (csrc/layer_rp.hip)


def _layer_fused(x, csr, edge_stages, node_stages, training, owner=None, gen=0, pack16=None, pack_only=False):
    """edge stage + per-target sum + two node stages in ONE launch (gsn_layer_fused_fwd_hip); None if the layer does not
    fit (shape, activation, or a BatchNorm1d that needs batch statistics)."""
    if not FUSED_LAYER or len(edge_stages) != 1 or len(node_stages) != 2:
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
    if pack16 is not None and PACK16_LAYER and L.gsn_layer_fused_pack16_supported(ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1)):
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
        out = torch.empty((n, node_stages[1].weight.shape[0]), dtype=torch.float32, device=x.device)
        pk = _abi.gsn_pack16()
        pk.node_rows = pack16[0].data_ptr()
        pk.edge_rows = None if pack16[1] is None else pack16[1].data_ptr()
        e_rows = 0 if pack16[1] is None else pack16[1].shape[0]
        with _abi.device_guard(x.device), _timed("layer_fused", flops):
            rc = L.gsn_layer_fused_fwd_pack16_hip(n, E, csr.seg_ptr.data_ptr(), ctypes.byref(ge), x.data_ptr(), d_x, ctypes.byref(g0), ctypes.byref(g1),
                                                  prep.data_ptr(), ctypes.byref(pk), e_rows, out.data_ptr(), _abi.current_stream())
        if rc != -2:                      # (GSN_E_UNSUPPORTED: packs beyond 32-bit offsets -> the fp32 kernel below)
            _abi.check(rc, "gsn_layer_fused_fwd_pack16_hip")
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
    if (GRAPH_ALIGNED_LAYER and part is not None and d_x == 128 and part[2] <= 128 and int(part[0].numel()) > 1
            and L.gsn_layer_fused_graphs_supported(ctypes.byref(ge), d_x, ctypes.byref(g0), ctypes.byref(g1))):
        with _abi.device_guard(x.device), _timed("layer_fused", flops):
            rc = L.gsn_layer_fused_fwd_graphs_hip(n, E, csr.seg_ptr.data_ptr(), ctypes.byref(ge), x.data_ptr(), d_x, ctypes.byref(g0),
                                                  ctypes.byref(g1), prep.data_ptr(), int(part[0].numel()) - 1, part[0].data_ptr(), int(part[2]),
                                                  out.data_ptr(), _abi.current_stream())
        if rc != -2:
            _abi.check(rc, "gsn_layer_fused_fwd_graphs_hip")
            return out
    # layers of a d = 128 model hand the row exponents of their output to the next one (csrc/layer_w.hip takes its edge rows' scales from
    # them): kept on the output tensor together with its version counter, used only while the tensor is unchanged
    x_exp = None
    hit = getattr(x, "_gsn_row_exp", None)
    # (a write through `.data` does not move the version counter -- the caveat of every per-tensor cache here, INTEGRATION.md: the
    #  validation mode does not trust the tensor's exponents and lets the kernel make them again)
    if hit is not None and hit[1] == x._version and hit[0].numel() == n and hit[0].device == x.device and not VALIDATE_CACHES:
        x_exp = hit[0]
    # (asked of the d = 128 kernel only, which writes them with its rows; behind the other kernels they would cost a pass over the output)
    out_exp = torch.empty(n, dtype=torch.int32, device=x.device) if d_x == 128 and out.shape[1] == 128 and CHAIN_ROW_EXPONENTS else None
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


FUSE_BN_ACT_ROWS = int(os.environ.get("GSN_FUSE_BN_ACT_ROWS", "16384"))      # train-mode stages of at most this many rows: finalize + normalise in one launch


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
                if RAW_WRITTEN is not None:              # (... and a replay of the captured add must move it too)
                    RAW_WRITTEN.append(bn.num_batches_tracked)
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
                # (h, act code, out): the normalise + activate pass rides the same launch (FUSE_BN_ACT_ROWS: where a launch costs more than it)
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
            if RAW_WRITTEN is not None:
                RAW_WRITTEN.extend(touched)
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



# ------------------------------------------------------------------------------------------------------------------
# native backward of dense stage lists (inputs = plain row-major blocks): gsn_bn_act_bwd_hip, gsn_wgrad_hip and the
# forward linear kernel on W^T for the input gradient
# ------------------------------------------------------------------------------------------------------------------
NATIVE_DENSE_BACKWARD = True      # False: every mlp backward goes through the PyTorch twin (for comparison)
GATHER_CAT_TRAIN = os.environ.get("GSN_GATHER_CAT_TRAIN", "0") == "1"    # 1: training assembles the edge rows first (gsn_gather_cat_hip), as before r03


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
        for si, sp in enumerate(spec):
            w = next(it)
            b = next(it) if sp["has_bias"] else None
            bn = sp["bn"]
            gamma = beta = None
            if bn is not None and bn.affine:
                gamma, beta = next(it), next(it)
            blks = [(t, ix) for t, ix in zip(blocks0, idx0)] if si == 0 else [(y, None)]
            n_out = w.shape[0]
            bn_train = bn is not None and (bn.training or bn.running_mean is None)
            bn_affine_grad = bn is not None and bn.affine and (bn.weight.requires_grad or bn.bias.requires_grad)
            if bn is not None and not bn_train and not bn_affine_grad:
                # BatchNorm on its running statistics, gamma / beta frozen: a per-column affine map in the epilogue of the product
                st = _Stage(w, b, bn, sp["act"])
                _bn_resolve(st, None, m_rows, False)
                mean32, scale, shift = st.bn_params
                y = _linear_hip(blks, w, b, mean32, scale, shift, _ACT_CODE[sp["act"]], m_rows)
                saved += [y, _f32c(scale)]
                meta.append(("affine", len(saved) - 2))
            elif bn is not None:
                # batch statistics (train mode), or running statistics with gradients for gamma / beta: pre-BN rows materialised
                st = _Stage(w, b, bn, sp["act"])
                fused_act = False
                if bn_train:
                    stats = _zeros(2 * n_out, torch.float64, w.device).view(2, n_out)
                    h = _linear_hip(blks, w, b, None, None, None, 0, m_rows, out=True, stats=stats)
                    yy = torch.empty_like(h)
                    fused_act = 1 < m_rows <= FUSE_BN_ACT_ROWS
                    _bn_resolve(st, lambda: stats, m_rows, True, fuse_act=(h, _ACT_CODE[sp["act"]], yy) if fused_act else None)
                else:
                    h = _linear_hip(blks, w, b, None, None, None, 0, m_rows, out=True)
                    yy = torch.empty_like(h)
                    _bn_resolve(st, None, m_rows, False)
                mean32, scale, shift = st.bn_params
                vecs = [_f32c(v) for v in (mean32, scale, shift)]
                if not fused_act:
                    with _abi.device_guard(h.device), _timed("bn_act", 8.0 * h.numel()):
                        _abi.check(_abi.lib().gsn_bn_act_hip(m_rows, n_out, h.data_ptr(), vecs[0].data_ptr(), vecs[1].data_ptr(),
                                                             vecs[2].data_ptr(), _ACT_CODE[sp["act"]], yy.data_ptr(),
                                                             _abi.current_stream()), "gsn_bn_act_hip")
                invstd = st.bn_invstd
                saved += [h, yy, vecs[0], invstd.contiguous(), vecs[1], vecs[2]]
                meta.append(("bn" if bn_train else "bn_eval", len(saved) - 6))
                y = yy
            else:
                y = _linear_hip(blks, w, b, None, None, None, _ACT_CODE[sp["act"]], m_rows)
                saved += [y]
                meta.append(("plain", len(saved) - 1))
        ctx.spec, ctx.meta, ctx.m_rows = spec, meta, m_rows
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
            gh = torch.empty((m_rows, n_out), dtype=torch.float32, device=dev)
            act = _ACT_CODE[sp["act"]]
            with _abi.device_guard(dev), _timed("bn_act_bwd", 16.0 * m_rows * n_out):
                if kind in ("bn", "bn_eval"):
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
            _abi.check(rc, "gsn_bn_act_bwd_hip")
            # (fp64 column sums -> fp32 gradients: ONE conversion of the whole arena behind the loop, the gradients are its slices)
            o0 = o64 - 3 * n_out
            if kind in ("bn", "bn_eval") and "g_i" in ent:
                casts.append((ent["g_i"], o0 + 2 * n_out, n_out))
                casts.append((ent["beta_i"], o0 + n_out, n_out))
            if "b_i" in ent:
                casts.append((ent["b_i"], o0, n_out))
            # weight gradient
            xin = blocks0 if si == 0 else [saved[meta[si - 1][1] + (1 if meta[si - 1][0] in ("bn", "bn_eval") else 0)]]
            if ctx.needs_input_grad[1 + ent["w_i"]]:
                gw = z32[o32:o32 + n_out * k_total].view(n_out, k_total)
                o32 += n_out * k_total
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
            # input gradient
            need_x = si > 0 or any(ctx.needs_input_grad[1 + bi] for bi in range(len(blocks0)))
            if need_x:
                # gX = gH W: W read as its transpose.  The fp16x3 kernel prepares its planes from any strides; the bf16x6 kernel stages a
                # strided W with scalar loads -- fine where a launch costs more than the staging (small batches), a copy + float4 staging above
                wt = w.detach().t() if (w.shape[1] > LINEAR_F16X3_MIN_N or m_rows <= 16384) else _transposed(w)
                gx = _linear_hip([(gh, None)], wt, None, None, None, None, 0, m_rows)
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


FOLD_KERNEL = os.environ.get("GSN_FOLD_KERNEL", "1") != "0"      # A/B switch: the fold as tensor ops over the dense stages (~18 launches per layer and step)


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
    if not NATIVE_DENSE_BACKWARD or not stages or len(stages[0].blocks) > 5:
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


CODE_STATUS_CHECK = True   # read the out-of-range flag back after every code-gather launch (one host sync)


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
    if CODE_STATUS_CHECK and int(status.item()) != 0:
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


class mlp(nn.Module):
    """models_misc.mlp (models_misc.py:18-59): Linear -> [BatchNorm1d] -> activation ... -> Linear, same attribute
    names (``fc``, ``bn``) so state dicts are interchangeable; forward runs on the HIP dense stages."""

    def __init__(self, in_features, out_features, d_k, seed, activation="elu", batch_norm=False):
        super().__init__()
        self.in_features, self.out_features, self.d_k, self.seed = in_features, out_features, d_k, seed
        self.activation_name, self.batch_norm = activation, batch_norm
        fc, bn = [], []
        d_in = [in_features]
        d_k = d_k + [out_features]
        for i in range(len(d_k)):
            fc.append(nn.Linear(d_in[i], d_k[i], bias=True))
            d_in = d_in + [d_k[i]]
            if self.batch_norm and i != len(d_k) - 1:
                bn.append(nn.BatchNorm1d(d_k[i]))
        self.fc = nn.ModuleList(fc)
        self.bn = nn.ModuleList(bn)
        self.activation = choose_activation(activation)

    def stages(self, blocks, upto=None, first_weight=None, first_bias=None, post=None):
        """The mlp as a list of _Stage; ``blocks`` feed the first Linear (optionally with a replaced first weight).
        ``post = (BatchNorm1d or None, activation name)`` is applied to the output of the last Linear (the model's
        between-layer BatchNorm + activation, models_graph_classification.py:226-228, fused into the stage epilogue)."""
        n = len(self.fc) if upto is None else upto
        out = []
        for i in range(n):
            last = i == len(self.fc) - 1
            w = self.fc[i].weight if (i > 0 or first_weight is None) else first_weight
            b = self.fc[i].bias if (i > 0 or first_bias is None) else first_bias
            bn = self.bn[i] if (self.batch_norm and not last) else None
            act = "identity" if last else self.activation_name
            if last and post is not None:
                bn, act = post
            out.append(_Stage(w, b, bn, act, blocks if i == 0 else ()))
        return out

    def hip_forward(self, blocks, m_rows, upto=None, first_weight=None, first_bias=None, csr=None, post=None):
        stages = self.stages(blocks, upto, first_weight, first_bias, post)
        if post is not None and post[0] is not None and post[0].training != self.training:
            raise RuntimeError("mlp and the fused BatchNorm1d must be in the same train/eval mode")
        return run_stages(stages, m_rows, self.training, csr=csr)

    # -- differentiable PyTorch twin (used to back-propagate through the dense stages)
    def torch_forward(self, x, upto=None, post=None):
        n = len(self.fc) if upto is None else upto
        for i in range(n):
            x = self.fc[i](x)
            if i != len(self.fc) - 1:
                if self.batch_norm:
                    b = self.bn[i]
                    x = F.batch_norm(x, None if self.training else b.running_mean, None if self.training else b.running_var,
                                     b.weight, b.bias, self.training, 0.0, b.eps)
                x = self.activation(x)
            elif post is not None:
                b, act = post
                if b is not None:
                    x = F.batch_norm(x, None if b.training else b.running_mean, None if b.training else b.running_var,
                                     b.weight, b.bias, b.training, 0.0, b.eps)
                x = choose_activation(act)(x)
        return x

    def forward(self, x, post=None):
        _need_cuda(x, "mlp input")
        native = _dense_native_ok(self.stages([(x, None)], post=post))
        if torch.is_grad_enabled() and self.training and native:
            return run_stages_autograd(self.stages([(x, None)], post=post), x.shape[0], self.training)
        # eval mode: the fused forward; if a gradient is asked for after all, the stages are re-run on the kernels that keep what
        # their HIP adjoints need (no PyTorch twin: BatchNorm on running statistics is a per-column affine map there)
        extra = list(post[0].parameters()) if (post is not None and post[0] is not None) else []
        again = (lambda x_: run_stages_autograd(self.stages([(x_, None)], post=post), x_.shape[0], self.training)) if native and not self.training \
            else (lambda x_: self.torch_forward(x_, post=post))
        return _run(self, lambda: self.hip_forward([(x, None)], x.shape[0], post=post), again, [x], extra_params=extra,
                    native=native and not self.training)


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
    fn = _HipWithNativeBackward if (native and NATIVE_DENSE_BACKWARD) else _HipWithTorchBackward
    return fn.apply(hip_fn, torch_fn, len(inputs), *inputs, *params)


def run_linear_module(lin, x):
    """A lone ``nn.Linear`` on the HIP dense stage (DiscreteEmbedding('linear'), utils_graph_learning.py:63-65)."""
    _need_cuda(x, "linear input")
    stage = lambda: run_stages([_Stage(lin.weight, lin.bias, None, "identity", [(x, None)])], x.shape[0], False)
    if NATIVE_DENSE_BACKWARD:
        again = lambda x_: run_stages_autograd([_Stage(lin.weight, lin.bias, None, "identity", [(x_, None)])], x_.shape[0], False)
        return _run(lin, stage, again, [x], native=True)
    return _run(lin, stage, lambda x_: F.linear(x_, lin.weight, lin.bias), [x])


# ------------------------------------------------------------------------------------------------------------------
# central_encoder (utils_graph_learning.py:211-260): dummy "self loop" value of ids / edge features for GIN-style sums
# ------------------------------------------------------------------------------------------------------------------
class _SumEmbedding(nn.Module):
    """the reference's multi_embedding([1], d, aggr='sum') -> parameter name ``encoder.0.weight``"""

    def __init__(self, d_out):
        super().__init__()
        emb = nn.Embedding(1, d_out)
        torch.nn.init.xavier_uniform_(emb.weight.data)
        self.encoder = nn.ModuleList([emb])


class _DiscreteEmbeddingShim(nn.Module):
    def __init__(self, d_out):
        super().__init__()
        self.encoder = _SumEmbedding(d_out)


class central_encoder(nn.Module):
    def __init__(self, nb_encoder, d_ef, extend=True):
        super().__init__()
        self.extend, self.nb_encoder = extend, nb_encoder
        self.one_hot = "one_hot_encoder" in nb_encoder
        if self.one_hot:
            self.d_out = d_ef + 1 if extend else d_ef
        else:
            self.d_out = d_ef
            if extend:
                self.encoder = _DiscreteEmbeddingShim(d_ef)  # state key: encoder.encoder.encoder.0.weight

    def forward(self, x_nb, num_nodes):
        if self.one_hot and self.extend:
            x_nb = torch.cat((torch.zeros((x_nb.shape[0], 1), device=x_nb.device), x_nb), -1)
            x_central = torch.zeros((num_nodes, self.d_out), device=x_nb.device)
            x_central[:, 0] = 1.0
        elif (not self.one_hot) and self.extend:
            x_central = self.encoder.encoder.encoder[0].weight[0:1].expand(num_nodes, -1)
        else:
            x_central = torch.zeros((num_nodes, self.d_out), device=x_nb.device)
        return x_central, x_nb

    def central_row(self, device):
        """(the central value as ONE row [1][d_out], zero columns to put in front of the neighbours' block): what ``forward`` expands to
        num_nodes rows and concatenates -- gsn_propagate_self_fwd_hip takes it as a row stride of 0 and a column offset instead."""
        if (not self.one_hot) and self.extend:
            return self.encoder.encoder.encoder[0].weight[0:1], 0
        key = str(device)
        cache = self.__dict__.setdefault("_gsn_rows", {})
        if key not in cache:
            row = torch.zeros((1, self.d_out), device=device)
            if self.one_hot and self.extend:
                row[0, 0] = 1.0
            cache[key] = row
        return cache[key], (1 if (self.one_hot and self.extend) else 0)


# ------------------------------------------------------------------------------------------------------------------
# the layers
# ------------------------------------------------------------------------------------------------------------------
class _SparseLayer(nn.Module):
    """Shared implementation; subclasses fix (has_ids, has_ef, ogb)."""

    has_ids = True
    has_ef = False
    ogb = False

    def __init__(self, d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                 d_ef=None, d_id=None, id_scope=None, aggr="add", msg_kind=None, eps=0, train_eps=False,
                 flow="source_to_target", **kwargs):
        super().__init__()
        if msg_kind is None:
            msg_kind = "ogb" if self.ogb else "general"
        if not self.ogb:
            d_msg = d_in if d_msg is None else d_msg
        self.flow, self.aggr, self.msg_kind = flow, aggr, msg_kind
        if self.has_ids:
            self.id_scope = id_scope
        self.degree_as_tag, self.retain_features = degree_as_tag, retain_features
        if degree_as_tag:
            d_in = d_in + d_degree if retain_features else d_degree
        d_id = d_id if self.has_ids else 0
        d_ef = d_ef if self.has_ef else 0
        name = self.__class__.__name__
        if self.ogb:
            if msg_kind != "ogb":
                raise NotImplementedError("msg kind {} is not currently supported.".format(msg_kind))
            self._init_eps(eps, train_eps)
            update_input_dim = d_in
        elif msg_kind == "gin":
            if self.has_ef:
                self.central_node_edge_encoder = central_encoder(kwargs["edge_embedding"], d_ef, extend=kwargs["extend_dims"])
                d_ef = self.central_node_edge_encoder.d_out
            if self.has_ids and id_scope == "local":
                self.central_node_id_encoder = central_encoder(kwargs["id_embedding"], d_id, extend=kwargs["extend_dims"])
                d_id = self.central_node_id_encoder.d_out
            self._init_eps(eps, train_eps)
            self.msg_fn = None
            update_input_dim = d_in + d_id + d_ef
        elif msg_kind == "general":
            local = (not self.has_ids) or id_scope == "local"
            msg_input_dim = 2 * d_in + d_id + d_ef if local else 2 * (d_in + d_id) + d_ef
            self.msg_fn = mlp(msg_input_dim, d_msg, d_h, seed, activation_name, bn)
            update_input_dim = d_in + d_msg
        else:
            raise NotImplementedError("msg kind {} is not currently supported.".format(msg_kind))
        self.update_fn = mlp(update_input_dim, d_up, d_h, seed, activation_name, bn)
        del name

    def _init_eps(self, eps, train_eps):
        self.initial_eps = eps
        if train_eps:
            self.eps = torch.nn.Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer("eps", torch.Tensor([eps]))
        self.eps.data.fill_(self.initial_eps)

    # -- input preparation shared by both paths (reference: forward() prologue of every layer)
    def _prepare(self, x, kwargs):
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        if self.degree_as_tag:
            x = _dense(x)
        degrees = kwargs["degrees"]
        identifiers = kwargs["identifiers"] if self.has_ids or self.ogb else None
        if not self.has_ids:
            identifiers = None
        if degrees is not None:
            degrees = degrees.unsqueeze(-1) if degrees.dim() == 1 else degrees
        if self.degree_as_tag:
            x = torch.cat([x, degrees], -1) if self.retain_features else degrees
        ef = None
        if self.has_ef:
            ef = kwargs["edge_features"]
            ef = ef.unsqueeze(-1) if ef.dim() == 1 else ef
        return x, identifiers, ef

    def forward(self, x, edge_index, **kwargs):
        if self.aggr == "mean":
            raise NameError("name 'aggr_index' is not defined")  # the reference's 'mean' branch is dead code (GSN_sparse.py:148)
        if self.aggr != "add":
            raise NotImplementedError("Aggregation kind {} is not currently supported.".format(self.aggr))
        if self.msg_kind not in ("gin", "general", "ogb"):
            raise NotImplementedError("Message kind {} is not currently supported.".format(self.msg_kind))
        post = None
        if kwargs.get("post_bn") is not None or kwargs.get("post_act") is not None:
            post = (kwargs.get("post_bn"), kwargs.get("post_act") or "identity")
        x, ids, ef = self._prepare(x, kwargs)
        _need_cuda(x, "x")
        if VALIDATE_CACHES:           # (see _module_fingerprint: notices parameter writes that bypass tensor._version)
            fp = _module_fingerprint(self)
            if fp != getattr(self, "_gsn_fingerprint", None):
                invalidate_caches(self)
                self._gsn_fingerprint = fp
        elif ASYNC_VALIDATE and not self.training and not torch.cuda.is_current_stream_capturing():
            _async_validate(self)
        # Row counts of the per-edge / per-vertex inputs: the reference fails in torch.cat / indexing when they do not fit
        # (GSN_sparse.py:118-132); the kernels would read past the tensors instead.
        n_rows, n_cols_e = x.shape[0], edge_index.shape[1]
        ids_per_edge = self.has_ids and (self.id_scope == "local")
        if ids is not None and ids.shape[0] != (n_cols_e if ids_per_edge else n_rows):
            raise RuntimeError("identifiers: %d rows, expected %d (id_scope %r: one row per %s)" % (
                ids.shape[0], n_cols_e if ids_per_edge else n_rows, self.id_scope, "edge" if ids_per_edge else "vertex"))
        if ef is not None and ef.shape[0] != n_cols_e:
            raise RuntimeError("edge_features: %d rows, expected one per edge (%d)" % (ef.shape[0], n_cols_e))
        _need_cuda(edge_index, "edge_index")
        given = [x, ids, ef]
        inputs = [t for t in given if t is not None and not isinstance(t, Codes)]

        def unpack(ts):     # float inputs come back from autograd, Codes are densified for the differentiable twin
            ts = list(ts)
            return tuple(None if t is None else (t.dense() if isinstance(t, Codes) else ts.pop(0)) for t in given)

        if torch.is_grad_enabled() and NATIVE_DENSE_BACKWARD and self.training:
            # training: compositions of kernels that each have a HIP adjoint -- no PyTorch twin
            if self.ogb or self.msg_kind == "gin":   # propagate -> axpy -> update_fn
                return self._twin(edge_index, _dense(x), _dense(ids), _dense(ef), post=post, native=True)
            if self._general_native_ok():
                return self._general_train(edge_index, _dense(x), _dense(ids), _dense(ef), post)
        extra = list(post[0].parameters()) if (post is not None and post[0] is not None) else []
        if NATIVE_DENSE_BACKWARD and not self.training:
            # eval mode: the fused forward keeps nothing; a gradient asked for after all re-runs the layer as the composition of kernels
            # that have HIP adjoints (the train-mode paths above; every BatchNorm1d in eval mode is an affine map there) -- no twin
            if self.ogb or self.msg_kind == "gin":
                again = lambda *ts: self._twin(edge_index, *unpack(ts), post=post, native=True)
            else:
                again = lambda *ts: self._general_train(edge_index, *unpack(ts), post)
            return _run(self, lambda: self._hip(edge_index, x, ids, ef, post), again, inputs, extra_params=extra, native=True)
        return _run(self, lambda: self._hip(edge_index, x, ids, ef, post), lambda *ts: self._twin(edge_index, *unpack(ts), post=post), inputs,
                    extra_params=extra)

    # -- message blocks ------------------------------------------------------------------------------------------
    def _sel(self):
        return 0 if self.flow == "target_to_source" else 1

    def _gin_parts(self, x, ids, ef, n):
        """(self parts, neighbour ids, neighbour ef, ids per node?) for the gin formulation"""
        self_parts = [x]
        ids_nb, ids_per_node = None, False
        if self.has_ids:
            if self.id_scope == "global":
                self_parts.append(ids); ids_nb, ids_per_node = ids, True
            else:
                c, ids_nb = self.central_node_id_encoder(ids, n)
                self_parts.append(c)
        ef_nb = None
        if self.has_ef:
            c, ef_nb = self.central_node_edge_encoder(ef, n)
            self_parts.append(c)
        return self_parts, ids_nb, ef_nb, ids_per_node

    def _self_plus_messages(self, edge_index, x, ids, ef):
        """(1 + eps) * self + sum of messages of the gin / ogb layers in ONE pass of the propagate kernel (forward: no elementwise
        tensor op, no concatenation; GSN_sparse.py:157-163, GSN_edge_sparse.py:95-109, GSN_edge_sparse_ogb.py:63-84 / :103-106)."""
        n, sel = x.shape[0], self._sel()
        if self.ogb:
            per_node = self.has_ids and self.id_scope == "global"
            return propagate(1, edge_index, sel, n, a=x, b=ids if self.has_ids else None, c=ef, b_per_node=per_node,
                             selfs=[x, ids] if per_node else [x], eps=self.eps)
        selfs, ids_nb, ef_nb, per_node, pads = [x], None, None, False, [0, 0]
        if self.has_ids:
            if self.id_scope == "global":
                selfs.append(ids); ids_nb, per_node = ids, True
            else:
                row, pads[0] = self.central_node_id_encoder.central_row(x.device)
                selfs.append(row); ids_nb = ids
        if self.has_ef:
            row, pad = self.central_node_edge_encoder.central_row(x.device)
            selfs.append(row); ef_nb = ef
            if ids_nb is None:      # (the edge features are then the kernel's block b)
                ids_nb, ef_nb, pads[0] = ef, None, pad
            else:
                pads[1] = pad
        return propagate(0, edge_index, sel, n, a=x, b=ids_nb, c=ef_nb, b_per_node=per_node, selfs=selfs, eps=self.eps, pads=pads)

    # -- HIP forward ---------------------------------------------------------------------------------------------
    def _hip(self, edge_index, x, ids, ef, post=None):
        n = x.shape[0]
        sel = self._sel()
        E = edge_index.shape[1]
        raw = (x, ids, ef)
        use_codes = (not self.ogb and self.msg_kind == "general" and len(self.msg_fn.fc) == 2 and E > 0
                     and all(isinstance(t, Codes) for t in raw if t is not None))
        # integer codes in, one launch: the one-hot encodings go straight into exact fp16 row packs (gsn_one_hot_pack16_hip: 64 / 32 bytes
        # per row, no fp32 one-hot tensor at all) and the layer runs on them (csrc/layer_rp.hip).  Identifiers may also be the tagged rows
        # of the counting kernel (gsn_amd.counting.count_batch(encoded_pack=...)).
        if (PACK16_LAYER and FUSED_LAYER and isinstance(raw[0], Codes) and not self.ogb and self.msg_kind == "general" and len(self.msg_fn.fc) == 2
                and E > 0 and not self.training and not (self.has_ids and self.id_scope != "local")
                and (raw[2] is None or isinstance(raw[2], Codes)) and (raw[1] is None or isinstance(raw[1], (Codes, torch.Tensor)))):
            y = self._fused_on_code_packs(edge_index, raw, n, E, sel, post)
            if y is not None:
                return y
        x = _f32c(_dense(x))
        if not use_codes:
            ids, ef = _dense(ids), _dense(ef)
        if self.ogb or self.msg_kind == "gin":
            xin = self._self_plus_messages(edge_index, x, ids, ef)
            return self.update_fn.hip_forward([(xin, None)], n, post=post)
        # general
        idx_i, idx_j = edge_index[sel].contiguous(), edge_index[1 - sel].contiguous()
        blocks = [(x, idx_i), (x, idx_j)]
        if self.has_ids:
            blocks += [(ids, None)] if self.id_scope == "local" else [(ids, idx_i), (ids, idx_j)]
        if self.has_ef:
            blocks.append((ef, None))
        mf = self.msg_fn
        uf = self.update_fn
        if len(mf.fc) >= 2:
            # all stages but the last Linear of msg_fn on E rows, then the sum aggregation S = sum_e r_e; the last Linear
            # commutes with the sum:  agg = W2 S + deg*b2, and it feeds update_fn's first Linear (weights [W3x | W3a]):
            #     [x | agg] W3^T = x W3x^T + S (W3a W2)^T + deg (W3a b2)^T
            # so it is folded into that Linear's weight (a [d_h x d_msg] by [d_msg x d_h] product, once per call).
            csr = _csr_for(edge_index, sel, n)
            s_agg = None
            if use_codes:
                # every input of msg_fn's first Linear is a one-hot code column: weight-row gather, no dense one-hot
                cblocks = [(raw[0], csr.tgt), (raw[0], csr.src)]
                if self.has_ids:
                    cblocks += [(raw[1], csr.perm)] if self.id_scope == "local" else [(raw[1], csr.tgt), (raw[1], csr.src)]
                if self.has_ef:
                    cblocks.append((raw[2], csr.perm))
                s_agg = _code_stage_segsum(mf, cblocks, csr, E)
                if s_agg is None:
                    ids, ef = _dense(ids), _dense(ef)
                    blocks = [(x, idx_i), (x, idx_j)]
                    if self.has_ids:
                        blocks += [(ids, None)] if self.id_scope == "local" else [(ids, idx_i), (ids, idx_j)]
                    if self.has_ef:
                        blocks.append((ef, None))
            # fused path: rows walked in target-sorted order; every block is gathered through ONE int32 index
            # (x_i: sorted target, x_j: sorted source, per-edge rows: perm), the scatter-add happens in the epilogue
            if s_agg is None and E > 0:
                sblocks = [(x, csr.tgt), (x, csr.src)]
                if self.has_ids:
                    sblocks += [(ids, csr.perm)] if self.id_scope == "local" else [(ids, csr.tgt), (ids, csr.src)]
                if self.has_ef:
                    sblocks.append((ef, csr.perm))
                # the whole layer in one launch where it fits (eval-mode BatchNorm, K_edge <= 80, widths <= 128)
                if post is None or post[0] is None or post[0].training == uf.training:
                    pk = None
                    if PACK16_LAYER and not (self.has_ids and self.id_scope != "local"):
                        # exact fp16 packs of the inputs, when their producers left them (gsn_amd.packs): looked up on the caller's tensors
                        pk = packs.lookup(raw[0], [t for t in (raw[1] if self.has_ids else None, raw[2] if self.has_ef else None) if t is not None])
                    y = _layer_fused(x, csr, mf.stages(sblocks, upto=len(mf.fc) - 1),
                                     uf.stages([(x, None)], first_weight=self._folded_first_weight(x.shape[1]), post=post),
                                     self.training, owner=self, gen=getattr(self, "_fold_gen", 0), pack16=pk)
                    if y is not None:
                        return y
                # wide edge rows (K > 160: layers 1.. of a d = 128 model, K = 260): the node part of the Linear once per NODE,
                # the rest as a gather-add inside the scatter kernel
                s_agg = self._split_edge_stage(x, ids, ef, csr, n, E)
                if s_agg is None:
                    s_agg = mf.hip_forward(sblocks, E, upto=len(mf.fc) - 1, csr=csr)
            if s_agg is None:
                r = mf.hip_forward(blocks, E, upto=len(mf.fc) - 1)
                s_agg = propagate(0, edge_index, sel, n, b=r)
            w_first = self._folded_first_weight(x.shape[1])
            return uf.hip_forward([(x, None), (s_agg, None), (csr.deg4, None)], n, first_weight=w_first, post=post)
        msgs = mf.hip_forward(blocks, E)
        agg = propagate(0, edge_index, sel, n, b=msgs)
        return uf.hip_forward([(x, None), (agg, None)], n, post=post)


    def _fused_on_code_packs(self, edge_index, raw, n, E, sel, post):
        """The one-launch layer fed from integer codes (see _hip); None when the shapes are outside the packed-row kernel."""
        mf, uf = self.msg_fn, self.update_fn
        if post is not None and post[0] is not None and post[0].training != uf.training:
            return None
        per_edge = [t for t in (raw[1] if self.has_ids else None, raw[2] if self.has_ef else None) if t is not None]
        d_x = sum(raw[0].n_classes)
        if d_x > packs.NODE_COLS - 4 or sum(packs._width(t) for t in per_edge) > packs.EDGE_COLS:
            return None
        pk = packs.from_codes(raw[0], per_edge)
        if pk is None:
            return None
        csr = _csr_for(edge_index, sel, n)
        dev = edge_index.device
        # (shapes and block identities for the stage descriptors: the packed-row kernel does not read the fp32 pointers -- no kernel runs here)
        xph = torch.empty((n, d_x), dtype=torch.float32, device=dev)
        sblocks = [(xph, csr.tgt), (xph, csr.src)]
        for t in per_edge:
            sblocks.append((t if isinstance(t, torch.Tensor) else torch.empty((E, packs._width(t)), dtype=torch.float32, device=dev), csr.perm))
        return _layer_fused(xph, csr, mf.stages(sblocks, upto=len(mf.fc) - 1),
                            uf.stages([(xph, None)], first_weight=self._folded_first_weight(d_x), post=post),
                            self.training, owner=self, gen=getattr(self, "_fold_gen", 0), pack16=pk, pack_only=True)

    # -- differentiable `general` path on native adjoints ------------------------------------------------------------------
    def _general_native_ok(self):
        return self.training

    def _general_train(self, edge_index, x, ids, ef, post):
        """msg_fn's hidden stages on materialised edge rows -> sum per target -> [x | S | deg] through update_fn with the
        folded first weight (the fold itself is three tiny PyTorch matrix products, so it stays differentiable)."""
        n, sel = x.shape[0], self._sel()
        E = edge_index.shape[1]
        tensors, modes = [x, x], [sel, 1 - sel]
        if self.has_ids:
            if self.id_scope == "local":
                tensors.append(ids); modes.append(None)
            else:
                tensors += [ids, ids]; modes += [sel, 1 - sel]
        if self.has_ef:
            tensors.append(ef); modes.append(None)
        mf, uf = self.msg_fn, self.update_fn
        # (an edge-less batch walks the same graph with zero rows: every parameter and input then gets the ZERO gradient PyTorch gives it)
        # the edge rows cat(x_i, x_j, ids.., e) are read where they lie (gathered blocks), by the product and by its weight gradient
        eblocks, gather = [(t, None) for t in tensors], (edge_index, n, tuple(modes))
        if GATHER_CAT_TRAIN or len(tensors) > _MAX_BLOCKS:      # (the assembled-rows form: A/B switch)
            eblocks, gather = [(_GatherCatFn.apply(edge_index, n, tuple(modes), *tensors), None)], None
        if len(mf.fc) < 2:      # a single Linear as msg_fn: nothing to fold (GSN_sparse.py:166-171 with d_h = [])
            msgs = run_stages_autograd(mf.stages(eblocks), E, True, gather=gather)
            agg = propagate(0, edge_index, sel, n, b=msgs)
            return run_stages_autograd(uf.stages([(x, None), (agg, None)], post=post), n, True)
        r = run_stages_autograd(mf.stages(eblocks, upto=len(mf.fc) - 1), E, True, gather=gather)
        s_agg = propagate(0, edge_index, sel, n, b=r)
        csr = _csr_for(edge_index, sel, n)
        last, w3 = mf.fc[-1], uf.fc[0].weight
        d_x = x.shape[1]
        if FOLD_KERNEL and _FoldWeightsFn.takes(w3, last, d_x):
            # the fold  W3x | W3a W2 | W3a b2  and its adjoint: one launch each (gsn_fold_weights_{fwd,bwd}_hip)
            # (three zero columns behind the degree column: the degree block goes in four floats wide, csr.deg4, and the stage's rows are staged
            #  as float4 -- K = d_x + d_h + 1 is odd otherwise)
            w_first = _FoldWeightsFn.apply(w3, last.weight, last.bias, d_x, 3)
            stages = uf.stages([(x, None), (s_agg, None), (csr.deg4, None)], first_weight=w_first, post=post)
            return run_stages_autograd(stages, n, True)
        else:
            # ... as two dense stages with their own adjoints (rows = W3a; no library GEMM in the step)
            w3x, w3a = w3[:, :d_x], w3[:, d_x:].contiguous()
            w_fold = run_stages_autograd([_Stage(last.weight.t(), None, None, "identity", [(w3a, None)])], w3a.shape[0], True)
            b_fold = run_stages_autograd([_Stage(last.bias.unsqueeze(0), None, None, "identity", [(w3a, None)])], w3a.shape[0], True)
            w_first = torch.cat([w3x, w_fold, b_fold], 1)
        stages = uf.stages([(x, None), (s_agg, None), (csr.deg, None)], first_weight=w_first, post=post)
        return run_stages_autograd(stages, n, True)

    def _split_edge_stage(self, x, ids, ef, csr, n, E):
        """S[t] = sum_{e -> t} act(bn(cat(x_i, x_j, z_e) W1^T + b1)) with the node columns of W1 applied once per node:
        P = x [W_i | W_j]^T (gsn_linear_fwd_hip, N rows), then gsn_edge_split_sum_hip gathers P_i[t] + P_j[src] and adds
        z_e W_z^T per edge.  Eval-mode / no BatchNorm, one hidden edge stage, identity / relu, per-edge blocks of
        <= 16 columns; used when the concatenated row is wider than the fused chain kernels take (K > SPLIT_EDGE_MIN_K).
        Returns None when the shape is outside that."""
        mf = self.msg_fn
        if not SPLIT_EDGE_STAGE or len(mf.fc) != 2 or self.training or E == 0:
            return None
        st = mf.stages([], upto=1)[0]
        act = {"identity": 0, "relu": 1}.get(st.act)
        if act is None or (st.bn is not None and (st.bn.training or st.bn.running_mean is None)):
            return None
        d_x, d_h = x.shape[1], st.weight.shape[0]
        node_blocks, edge_blocks = [x], []
        if self.has_ids:
            (edge_blocks if self.id_scope == "local" else node_blocks).append(ids)
        if self.has_ef:
            edge_blocks.append(ef)
        d_n = sum(b.shape[1] for b in node_blocks)
        d_r = sum(b.shape[1] for b in edge_blocks)
        if 2 * d_n + d_r <= SPLIT_EDGE_MIN_K or d_h % 4 or d_h > 256 or any(b.shape[1] % 4 for b in edge_blocks) \
                or d_r > (16 if d_h <= 128 else 8) or len(edge_blocks) > 2:
            return None
        _bn_resolve(st, None, E, False)
        w1, b1 = mf.fc[0].weight, mf.fc[0].bias
        bn_key = None if st.bn is None else tuple(t._version for t in (st.bn.running_mean, st.bn.running_var)) + \
            ((st.bn.weight._version, st.bn.bias._version) if st.bn.affine else ())
        key = (w1._version, w1.data_ptr(), None if b1 is None else b1._version, bn_key, d_x, d_n, d_r)
        cache = getattr(self, "_split_cache", None)
        if cache is None or cache[0] != key:
            w = w1.detach()
            scale = None
            bias = b1.detach() if b1 is not None else torch.zeros(d_h, device=w.device)
            if st.bn_params is not None:
                mean, scale, shift = st.bn_params
                bias = (bias - mean) * scale + shift
            # column layout of W1: x_i, x_j, then ids_i, ids_j (global scope) or ids (local), then edge features
            cols_i, cols_j, off = [w[:, :d_x]], [w[:, d_x:2 * d_x]], 2 * d_x
            if self.has_ids and self.id_scope != "local":
                d_id = ids.shape[1]
                cols_i.append(w[:, off:off + d_id]); cols_j.append(w[:, off + d_id:off + 2 * d_id]); off += 2 * d_id
            w_i, w_j, w_z = torch.cat(cols_i, 1), torch.cat(cols_j, 1), w[:, off:]
            if scale is not None:
                w_i, w_j, w_z = w_i * scale[:, None], w_j * scale[:, None], w_z * scale[:, None]
            w_n = torch.cat([w_i, w_j], 0).contiguous()                                   # [2 d_h, d_n]
            bias_n = torch.cat([bias, torch.zeros_like(bias)]).contiguous()               # the target half carries bias + BN shift
            wz_t = w_z.t().contiguous() if d_r else None                                  # [d_r, d_h]
            cache = (key, w_n, bias_n, wz_t)
            self._split_cache = cache
            _note_cache(self, "_split_cache")
        _, w_n, bias_n, wz_t = cache
        P = _linear_hip([(b, None) for b in node_blocks], w_n, bias_n, None, None, None, 0, n)          # [N, 2 d_h]
        zs = [_f32c(b) for b in edge_blocks]
        out = torch.empty((n, d_h), dtype=torch.float32, device=x.device)
        with _abi.device_guard(x.device), _timed("edge_split_sum", 4.0 * (E * (d_h + d_r) + 2.0 * n * d_h)):
            rc = _abi.lib().gsn_edge_split_sum_hip(n, E, csr.seg_ptr.data_ptr(), csr.src.data_ptr(), csr.perm.data_ptr(),
                                                   P.data_ptr(), P.data_ptr() + 4 * d_h, 2 * d_h,
                                                   zs[0].data_ptr() if zs else None, zs[0].shape[1] if zs else 0,
                                                   zs[1].data_ptr() if len(zs) > 1 else None, zs[1].shape[1] if len(zs) > 1 else 0,
                                                   _abi.ptr(wz_t), d_h, act, out.data_ptr(), _abi.current_stream())
        _abi.check(rc, "gsn_edge_split_sum_hip")
        return out

    def _folded_first_weight(self, d_x):
        """[W3x | W3a W2 | W3a b2] (see _hip); recomputed only when one of the three parameters changed."""
        mf, uf = self.msg_fn, self.update_fn
        last, w3p = mf.fc[-1], uf.fc[0].weight
        key = (last.weight._version, last.bias._version, w3p._version, last.weight.data_ptr(), w3p.data_ptr(), d_x)
        cache = getattr(self, "_fold_cache", None)
        if cache is not None and cache[0] == key:
            return cache[1]
        w3 = w3p.detach()
        w3x, w3a = w3[:, :d_x], w3[:, d_x:].contiguous()
        w2t = last.weight.detach().t().contiguous()                       # [d_h_msg, d_msg]: rows = input index
        w_fold = _linear_hip([(w3a, None)], w2t, None, None, None, None, 0, w3a.shape[0])   # = W3a @ W2
        b_fold = _linear_hip([(w3a, None)], last.bias.detach().unsqueeze(0).contiguous(), None, None, None, None, 0, w3a.shape[0])
        # three zero columns after the degree column: the degree block is passed 4 floats wide (csr.deg4)
        w_first = torch.cat([w3x, w_fold, b_fold, torch.zeros_like(b_fold).expand(-1, 3)], 1).contiguous()
        self._fold_cache = (key, w_first)
        _note_cache(self, "_fold_cache")
        self._fold_gen = getattr(self, "_fold_gen", 0) + 1        # (a new tensor may reuse the old one's address)
        return w_first

    # -- differentiable twin (PyTorch ops + the HIP propagate with its own adjoint) ---------------------------------
    def _twin(self, edge_index, x, ids, ef, post=None, native=False):
        n = x.shape[0]
        sel = self._sel()
        if native and (self.ogb or self.msg_kind == "gin"):
            return self.update_fn(self._self_plus_messages(edge_index, x, ids, ef), post=post)
        if self.ogb:
            per_node = self.has_ids and self.id_scope == "global"
            agg = propagate(1, edge_index, sel, n, a=x, b=ids if self.has_ids else None, c=ef, b_per_node=per_node)
            self_msg = x + ids if per_node else x
            xin = (1 + self.eps) * self_msg + agg
            return self.update_fn(xin, post=post) if native else self.update_fn.torch_forward(xin, post=post)
        if self.msg_kind == "gin":
            self_parts, ids_nb, ef_nb, per_node = self._gin_parts(x, ids, ef, n)
            agg = propagate(0, edge_index, sel, n, a=x, b=ids_nb, c=ef_nb, b_per_node=per_node)
            xin = (1 + self.eps) * torch.cat(self_parts, -1) + agg
            return self.update_fn(xin, post=post) if native else self.update_fn.torch_forward(xin, post=post)
        idx_i, idx_j = edge_index[sel], edge_index[1 - sel]
        parts = [x[idx_i], x[idx_j]]
        if self.has_ids:
            parts += [ids] if self.id_scope == "local" else [ids[idx_i], ids[idx_j]]
        if self.has_ef:
            parts.append(ef)
        msgs = self.msg_fn.torch_forward(torch.cat(parts, -1))
        agg = propagate(0, edge_index, sel, n, b=msgs)
        return self.update_fn.torch_forward(torch.cat((x, agg), -1), post=post)

    def __repr__(self):
        if self.ogb:
            return "{}(update_fn = {})".format(self.__class__.__name__, self.update_fn)
        return "{}(msg_fn = {}, update_fn = {})".format(self.__class__.__name__, self.msg_fn, self.update_fn)


class GSN_sparse(_SparseLayer):
    """graph_filters/GSN_sparse.py:8-181 (GSN-v: id_scope='global', GSN-e: 'local'; no edge features)."""
    has_ids, has_ef, ogb = True, False, False

    def __init__(self, d_in, d_id, d_degree, degree_as_tag, retain_features, id_scope, d_msg, d_up, d_h, seed,
                 activation_name, bn, aggr="add", msg_kind="general", eps=0, train_eps=False, flow="source_to_target", **kwargs):
        kwargs.pop("d_ef", None)
        super().__init__(d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                         d_id=d_id, id_scope=id_scope, aggr=aggr, msg_kind=msg_kind, eps=eps, train_eps=train_eps, flow=flow, **kwargs)


class GSN_edge_sparse(_SparseLayer):
    """graph_filters/GSN_edge_sparse.py:8-175."""
    has_ids, has_ef, ogb = True, True, False

    def __init__(self, d_in, d_ef, d_id, d_degree, degree_as_tag, retain_features, id_scope, d_msg, d_up, d_h, seed,
                 activation_name, bn, aggr="add", msg_kind="general", eps=0, train_eps=False, flow="source_to_target", **kwargs):
        super().__init__(d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                         d_ef=d_ef, d_id=d_id, id_scope=id_scope, aggr=aggr, msg_kind=msg_kind, eps=eps, train_eps=train_eps,
                         flow=flow, **kwargs)


class MPNN_sparse(_SparseLayer):
    """graph_filters/MPNN_sparse.py (identifier-free twin of GSN_sparse)."""
    has_ids, has_ef, ogb = False, False, False

    def __init__(self, d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                 aggr="add", msg_kind="general", eps=0, train_eps=False, flow="source_to_target", **kwargs):
        kwargs.pop("d_ef", None)
        super().__init__(d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                         aggr=aggr, msg_kind=msg_kind, eps=eps, train_eps=train_eps, flow=flow, **kwargs)


class MPNN_edge_sparse(_SparseLayer):
    """graph_filters/MPNN_edge_sparse.py (identifier-free twin of GSN_edge_sparse)."""
    has_ids, has_ef, ogb = False, True, False

    def __init__(self, d_in, d_ef, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                 aggr="add", msg_kind="general", eps=0, train_eps=False, flow="source_to_target", **kwargs):
        super().__init__(d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                         d_ef=d_ef, aggr=aggr, msg_kind=msg_kind, eps=eps, train_eps=train_eps, flow=flow, **kwargs)


class GSN_edge_sparse_ogb(_SparseLayer):
    """graph_filters/GSN_edge_sparse_ogb.py:9-134: msg = relu(x_j + id + e), GIN-style update."""
    has_ids, has_ef, ogb = True, True, True

    def __init__(self, d_in, d_ef, d_id, d_degree, degree_as_tag, retain_features, id_scope, d_msg, d_up, d_h, seed,
                 activation_name, bn, aggr="add", msg_kind="ogb", eps=0, train_eps=False, flow="source_to_target", **kwargs):
        super().__init__(d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                         d_ef=d_ef, d_id=d_id, id_scope=id_scope, aggr=aggr, msg_kind=msg_kind, eps=eps, train_eps=train_eps,
                         flow=flow, **kwargs)


class MPNN_edge_sparse_ogb(_SparseLayer):
    """graph_filters/MPNN_edge_sparse_ogb.py: msg = relu(x_j + e)."""
    has_ids, has_ef, ogb = False, True, True

    def __init__(self, d_in, d_ef, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                 aggr="add", msg_kind="ogb", eps=0, train_eps=False, flow="source_to_target", **kwargs):
        super().__init__(d_in, d_degree, degree_as_tag, retain_features, d_msg, d_up, d_h, seed, activation_name, bn,
                         d_ef=d_ef, aggr=aggr, msg_kind=msg_kind, eps=eps, train_eps=train_eps, flow=flow, **kwargs)

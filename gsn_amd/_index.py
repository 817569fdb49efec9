"""Index side of the layers: integer-coded inputs (``Codes``), the target-sorted CSR of an aggregation index and its caches (keyed on the
input tensors), registered graph / batch partitions, the segment-sum readouts and the scatter-add ``propagate`` with its HIP adjoint
(utils_graph_learning.py:170-260, GSN_sparse.py:150-163)."""
from __future__ import annotations

import weakref

import torch

from . import _abi, flags
from ._runtime import INPUT_EPOCH, _f32c, _need_cuda, _timed, _zeros

class Codes:
    """Integer category codes standing in for their one-hot encoding (the output of the reference's
    DiscreteEmbedding('one_hot_encoder'), utils_graph_learning.py:170-187) as a layer input.

    ``codes`` int64 [R, C] on the GPU, ``n_classes`` C ints; equivalent to the float tensor ``dense()`` of shape
    [R, sum(n_classes)].  Layers accept it for ``x``, ``identifiers`` and ``edge_features``; where all inputs of msg_fn's
    first Linear are Codes that Linear becomes a weight-row gather (gsn_code_stage_fwd_hip) and the dense one-hot
    matrix is never built; everywhere else the layer densifies it."""
    __slots__ = ("codes", "n_classes", "clamp", "check", "_dense", "_pack16", "__weakref__")

    def __init__(self, codes, n_classes, clamp=False, check=None):
        codes = codes.unsqueeze(-1) if codes.dim() == 1 else codes
        _need_cuda(codes, "codes")
        self.codes = codes.to(torch.int64).contiguous()
        self.n_classes = [int(c) for c in n_classes]
        if len(self.n_classes) != self.codes.shape[1]:
            raise ValueError("Codes: %d columns but %d class counts" % (self.codes.shape[1], len(self.n_classes)))
        self.clamp = bool(clamp)      # values above the last class count as the last class (as gsn_one_hot_hip's clamp)
        # read the out-of-range flag back after encoding / gathering these codes (a host synchronisation; IndexError as F.one_hot): None = the
        # process-wide flags.CODE_STATUS_CHECK, False = never (what the dense one_hot_encoder does: an out-of-range code gives a zero block)
        self.check = check
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
        if self._dense is None or self._dense[1] != self.codes._version or self._dense[2] != INPUT_EPOCH[0]:
            self._dense = (one_hot_identifiers(self.codes, self.n_classes, clamp=self.clamp), self.codes._version, INPUT_EPOCH[0])
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
    # the declared sizes are trusted by kernels that size their tiles by them (the graph-aligned d = 128 layer stops at 128 rows of a graph):
    # checked here when that is free (pointers still on the host, as at collate time) or asked for (``check``: one device read)
    if check or not node_ptr.is_cuda:
        for name, ptr, cap in (("node_ptr", node_ptr, max_nodes), ("edge_ptr", edge_ptr, max_edges)):
            if ptr.numel() > 1 and int((ptr[1:] - ptr[:-1]).max()) > int(cap):
                raise ValueError("set_graph_partition: a graph is larger than the declared max (%s: %d > %d)" % (name, int((ptr[1:] - ptr[:-1]).max()), int(cap)))
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
        # the ogb layers: out = (1 + eps) x + sum relu(x_j + ..) with the self block BEING the gathered block -- its adjoint rides the node pass
        # (gsn_propagate_bwd_fold_self_hip: g_a receives both contributions, g_eps its fp64 sum; no pass over [N d] for g_self, no
        # gradient-accumulation add of g_a + g_self behind this function)
        want_eps = ctx.has_eps and ctx.needs_input_grad[9]
        folded = False
        if (ctx.kind == 1 and ctx.n_self == 1 and not ctx.b_per_node and g_a is not None and E > 0 and (g_b is not None or g_c is not None)
                and a is not None and ss[0].data_ptr() == a.data_ptr() and ss[0].shape == a.shape and ctx.needs_input_grad[10]
                and ctx.pads == (0, 0) and flags.FOLD_SELF_ADJOINT):
            acc = _zeros(1, torch.float64, dev) if want_eps else None
            with _abi.device_guard(dev):
                rc = _abi.lib().gsn_propagate_bwd_fold_self_hip(n, E, src.data_ptr(), tgt.data_ptr(), csr_s.seg_ptr.data_ptr(), csr_s.perm.data_ptr(),
                                                                a.data_ptr(), wa, _abi.ptr(b), wb, _abi.ptr(c), wc, g_out.data_ptr(), g_a.data_ptr(),
                                                                _abi.ptr(g_b), _abi.ptr(g_c), eps32.data_ptr() if ctx.has_eps else None,
                                                                _abi.ptr(acc), _abi.current_stream())
            if rc != -2:             # (GSN_E_UNSUPPORTED: nothing was launched -- the two-function path below)
                _abi.check(rc, "gsn_propagate_bwd_fold_self_hip")
                folded = True
                if shared_bc:
                    g_b = g_c
                g_eps = acc[0].to(torch.float32).reshape(ctx.eps_shape) if want_eps else None
                return (None, None, None, None, None, g_a, g_b, g_c, None, g_eps, None)
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

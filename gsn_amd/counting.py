"""Substructure / orbit counting on the GPU: host-side mirror of the reference's counting interface.

* :func:`subgraph_isomorphism_vertex_counts` / :func:`subgraph_isomorphism_edge_counts` -- same signatures and return
  type (CPU float64 tensor) as utils_graph_processing.py:103-131 / :134-179; a G=1 call into the batched kernel.
* :func:`subgraph_counts2ids` -- same signature and side effects as utils_ids.py:7-29 (strips self loops, concatenates
  the per-pattern counts, writes ``data.edge_index`` and ``data.identifiers`` int64); all patterns go down in ONE
  kernel launch instead of one graph-tool call per pattern.
* :func:`count_batch` / :func:`counts2ids_batch` -- the batched driver: a whole dataset shard per launch.

Everything routes through ``gsn_count_hip`` in libgsn_hip.so.  No CPU fallback: without a GPU these raise.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _abi, packs
from .patterns import PatternGraph

__all__ = ["CountPlan", "count_batch", "counts2ids_batch", "subgraph_isomorphism_vertex_counts",
           "subgraph_isomorphism_edge_counts", "subgraph_counts2ids"]

GSN_KMAX = 9                                            # include/gsn_abi.h
PLAN_STRIDE_WORDS = 2 + GSN_KMAX + (GSN_KMAX + 3) // 4  # csrc/gsn_internal.h (undirected plans)
_MODE = {"vertex": 0, "edge": 1}
_STATUS_MSG = {2: "graph larger than the max_nodes / max_edges given to the call", 3: "vertex id outside [0, num_nodes)"}
_PLAN_CACHE = {}


class CountPlan:
    """Compiled search plans for a list of patterns (built on the host by gsn_count_plan_build, cached per device)."""

    def __init__(self, pattern_edge_lists, mode, induced, directed_orbits=False, directed=False):
        self.mode = mode
        self.induced = bool(induced)
        self.directed_orbits = bool(directed_orbits)
        self.directed = bool(directed)           # digraph patterns and targets (main.py --directed); vertex mode only
        pats = [np.asarray(list(el), dtype=np.int64).reshape(-1, 2) for el in pattern_edge_lists]
        if not pats:
            raise ValueError("no patterns given")
        pat_ptr = np.cumsum([0] + [len(p) for p in pats]).astype(np.int64)
        pat_edges = np.ascontiguousarray(np.concatenate(pats, axis=0))
        L = _abi.lib()
        words, ncols = ctypes.c_int64(), ctypes.c_int64()
        args = (_MODE[mode], int(self.induced), int(self.directed_orbits) | (2 if self.directed else 0), len(pats),
                _abi.ptr(pat_ptr), _abi.ptr(pat_edges))
        _abi.check(L.gsn_count_plan_build(*args, None, 0, ctypes.addressof(words), ctypes.addressof(ncols)),
                   "gsn_count_plan_build")
        self.table = np.zeros(words.value, dtype=np.uint32)
        _abi.check(L.gsn_count_plan_build(*args, _abi.ptr(self.table), len(self.table), ctypes.addressof(words),
                                          ctypes.addressof(ncols)), "gsn_count_plan_build")
        self.n_cols = int(ncols.value)
        self.n_plans = int(self.table[3])
        self.kmax = int(self.table[5])
        self._dev = {}

    @staticmethod
    def get(pattern_edge_lists, mode, induced, directed_orbits=False, directed=False):
        key = (tuple(tuple((int(u), int(v)) for u, v in el) for el in pattern_edge_lists), mode, bool(induced),
               bool(directed_orbits), bool(directed))
        if key not in _PLAN_CACHE:
            _PLAN_CACHE[key] = CountPlan(pattern_edge_lists, mode, induced, directed_orbits, directed)
        return _PLAN_CACHE[key]

    def device_table(self, device):
        device = torch.device(device)
        if device not in self._dev:
            self._dev[device] = torch.from_numpy(self.table.view(np.int32)).to(device)
        return self._dev[device]


def _as_dev_i64(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.int64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.int64), device=device)


def count_batch(plan: CountPlan, node_ptr, edge_ptr, edge_index, ids_are_global=True, max_nodes=None, max_edges=None,
                device=None, graph_ids=None, out=None, check=True, encode=None, counts=True, encoded_out=None, encoded_pack=None,
                encoded_rows=True):
    """Run the counting kernel over a batch.

    ``encode=(n_classes, clamp)`` also returns the one-hot encoded identifiers the GSN layers consume (the reference's
    ``DiscreteEmbedding('one_hot_encoder')`` over the counts, utils_graph_learning.py:170-187) straight from the kernel
    (``gsn_count_encode_hip``): the result becomes ``(out, status, encoded)`` with ``encoded`` fp32
    [rows_total, sum(n_classes)] (``encoded_out`` to reuse a buffer); with ``counts=False`` the int64 rows are not written at
    all (``out`` is None).  ``encoded_rows=False`` (with ``encoded_pack`` and ``counts=True``): NO fp32 rows are written -- the int64 counts and
    the pack's identifier columns leave the kernel, and ``encoded`` is a :class:`gsn_amd.layers.Codes` over the counts (clamped one-hot classes:
    what the columns hold) tagged with the pack: the layers take it as ``identifiers``.  ``encoded_pack=(pack, col0)`` (with ``encode``): the kernel writes the encoded rows a second time as
    fp16 into columns col0.. of the exact row pack ``pack`` (:mod:`gsn_amd.packs`) and ``encoded`` is tagged with it -- the first GSN
    layer then reads the pack (csrc/layer_rp.hip) instead of converting the fp32 rows.

    node_ptr / edge_ptr: int64 [G+1] (host or device); edge_index: int64 [2, E_total].  Returns
    ``(out, status)``: ``out`` int64 device tensor [rows_total, plan.n_cols] (rows = vertices in vertex mode, columns
    of edge_index in edge mode), ``status`` int32 [G].  With ``check`` the statuses are read back and the reference's
    errors are raised (KeyError when a match uses a direction that is not a column, utils_graph_processing.py:173).
    """
    _abi.require_gpu()
    if device is None:
        device = edge_index.device if isinstance(edge_index, torch.Tensor) and edge_index.is_cuda else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if max_nodes is None or max_edges is None:
        npt = node_ptr.cpu().numpy() if isinstance(node_ptr, torch.Tensor) else np.asarray(node_ptr)
        ept = edge_ptr.cpu().numpy() if isinstance(edge_ptr, torch.Tensor) else np.asarray(edge_ptr)
        max_nodes = int(np.diff(npt).max()) if len(npt) > 1 else 1
        max_edges = int(np.diff(ept).max()) if len(ept) > 1 else 0
    node_ptr_d = _as_dev_i64(node_ptr, device)
    edge_ptr_d = _as_dev_i64(edge_ptr, device)
    ei = _as_dev_i64(edge_index, device)
    if ei.dim() != 2 or ei.shape[0] != 2:
        raise ValueError("edge_index must be [2, E]")
    n_graphs = node_ptr_d.numel() - 1
    E_total = ei.shape[1]
    enc = enc_tab = None
    pack_only = pack_codes_later = False
    rows_total = None
    if encode is not None or out is None:
        # rows follow the pointers the kernel uses, not the tensor length (one host read; pass `out` / `encoded_out` to avoid it)
        if out is not None:
            rows_total = out.shape[0]
        elif encoded_out is not None:
            rows_total = encoded_out.shape[0]
        else:
            rows_total = int((edge_ptr_d if plan.mode == "edge" else node_ptr_d)[-1].item()) if n_graphs > 0 else 0
    if encode is not None:
        n_classes, clamp = encode
        n_classes = [int(c) for c in n_classes]
        if len(n_classes) != plan.n_cols or min(n_classes) < 1:
            raise ValueError("encode: one n_classes >= 1 per output column (%d columns)" % plan.n_cols)
        enc_tab = np.asarray(n_classes, dtype=np.int32)
        pack_only = not encoded_rows
        if pack_only and (encoded_pack is None or not counts or encoded_out is not None):
            raise ValueError("encoded_rows=False: the encoded identifiers leave as pack columns next to the int64 counts (encoded_pack and counts=True, no encoded_out)")
        if pack_only and graph_ids is not None:
            # (the Codes object returned for the layer covers EVERY row of `out` / the pack; rows of graphs outside the list would be torch.empty)
            raise ValueError("encoded_rows=False: every graph of the batch is counted (no graph_ids)")
        if not pack_only:
            enc = encoded_out if encoded_out is not None else torch.empty((rows_total, sum(n_classes)), dtype=torch.float32, device=device)
            if enc.shape != (rows_total, sum(n_classes)) or enc.dtype != torch.float32 or not enc.is_contiguous():
                raise ValueError("encoded_out must be a contiguous fp32 [rows, sum(n_classes)] tensor")
    if out is None and (counts or encode is None):
        out = torch.empty((rows_total, plan.n_cols), dtype=torch.int64, device=device)
    # (the library zeroes the status words of the graphs it counts; with a subset the others must read OK as well)
    status = (torch.zeros if graph_ids is not None or n_graphs == 0 else torch.empty)(max(n_graphs, 1), dtype=torch.int32, device=device)
    gid = None
    n_items = n_graphs
    if graph_ids is not None:
        gid = torch.as_tensor(graph_ids, dtype=torch.int32, device=device).contiguous()
        n_items = gid.numel()
    if n_graphs > 0 and n_items > 0:
        tab = plan.device_table(device)
        common = (_abi.ptr(plan.table), tab.data_ptr(), len(plan.table), n_graphs, node_ptr_d.data_ptr(), edge_ptr_d.data_ptr(),
                  ei.data_ptr() if E_total else None, ei.stride(0), int(bool(ids_are_global)), None if gid is None else gid.data_ptr(),
                  n_items, int(max_nodes), int(max_edges), None if out is None else out.data_ptr(), status.data_ptr())
        with _abi.device_guard(device):
            if encode is not None and pack_only:
                # the int64 counts + the pack's identifier columns; no fp32 one-hot rows.  The layer input is a Codes object over the counts
                # (clamped one-hot classes: exactly what the columns hold), tagged with the pack
                pack, col0 = encoded_pack
                if pack.dtype != torch.float16 or pack.dim() != 2 or pack.shape[0] != rows_total or not pack.is_contiguous() or pack.device != out.device:
                    raise ValueError("encoded_pack: a contiguous fp16 [rows, cols] tensor on the device of the counts")
                rc = _abi.lib().gsn_count_encode_pack16_hip(*common, _abi.ptr(enc_tab), int(bool(encode[1])), None, pack.data_ptr(),
                                                            pack.shape[1], int(col0), _abi.current_stream())
                if rc == -2:                         # GSN_E_UNSUPPORTED: rows not staged in this launch configuration -> counts, then the codes' own packer
                    rc = _abi.lib().gsn_count_hip(*common, _abi.current_stream())
                    pack_codes_later = True
            elif enc is None:
                rc = _abi.lib().gsn_count_hip(*common, _abi.current_stream())
            elif encoded_pack is None:
                packs.release(enc)                  # (rewritten through its raw pointer: an earlier pack no longer describes it)
                rc = _abi.lib().gsn_count_encode_hip(*common, _abi.ptr(enc_tab), int(bool(encode[1])), enc.data_ptr(), _abi.current_stream())
            else:
                pack, col0 = encoded_pack
                if pack.dtype != torch.float16 or pack.dim() != 2 or pack.shape[0] != enc.shape[0] or not pack.is_contiguous() or pack.device != enc.device:
                    raise ValueError("encoded_pack: a contiguous fp16 [rows, cols] tensor on the device of the encoded rows")
                packs.release(enc)
                rc = _abi.lib().gsn_count_encode_pack16_hip(*common, _abi.ptr(enc_tab), int(bool(encode[1])), enc.data_ptr(), pack.data_ptr(),
                                                            pack.shape[1], int(col0), _abi.current_stream())
                if rc == -2:                         # GSN_E_UNSUPPORTED: rows not staged in this launch configuration -> pack the fp32 rows
                    rc = _abi.lib().gsn_count_encode_hip(*common, _abi.ptr(enc_tab), int(bool(encode[1])), enc.data_ptr(), _abi.current_stream())
                    _abi.check(rc, "gsn_count_encode_hip")
                    packs._pack_rows(enc, pack, int(col0), -1, False)
                if rc == 0:
                    packs.claim(enc, pack, int(col0))
        _abi.check(rc, "gsn_count_hip" if enc is None else "gsn_count_encode_hip")
    if check:
        _raise_statuses(status)
    if encode is not None and pack_only:
        from ._index import Codes
        cd = Codes(out, n_classes, clamp=bool(clamp), check=False)
        if pack_codes_later:
            # (the same shape rules as the kernel path: any fp16 [rows, cols] pack, columns col0 .. col0 + sum(n_classes))
            packs._pack_codes(cd, encoded_pack[0], int(encoded_pack[1]), -1)
            packs._claim_codes(cd, encoded_pack[0], int(encoded_pack[1]))
        elif n_graphs > 0 and n_items > 0:
            packs._claim_codes(cd, encoded_pack[0], int(encoded_pack[1]))
        return out, status, cd
    if encode is not None:
        return out, status, enc
    return out, status


def count_batch_side(plan, node_ptr, edge_ptr, edge_index, max_nodes, max_edges, id_classes=None, clamp=True, x_codes=None, ef_codes=None,
                     csr_row=None, out=None, ids_are_global=True, register=True, n_nodes=None):
    """:func:`count_batch` whose workgroups also leave what the first GSN layer needs (``gsn_count_encode_pack16_side_hip``): the
    target-sorted CSR of ``edge_index[csr_row]`` (``csr_row`` 0 / 1, None: no CSR; GSN_sparse.py:140-143), the node pack of the integer
    vertex codes ``x_codes`` (:class:`gsn_amd.layers.Codes`; utils_graph_learning.py:170-187) and -- edge-mode plans with ``id_classes`` --
    the whole edge pack rows: the identifiers' one-hot classes next to the one-hot of the edge codes ``ef_codes``.

    node_ptr / edge_ptr int64 device [G + 1] of a collated batch (node_ptr[0] = 0), edge_index int64 device [2, E].  Returns a dict:
    ``ids`` int64 [rows, plan.n_cols], ``status`` int32 [G] (not read back), ``code_status`` int32 [1], ``csr`` (a layers-side CSR object
    or None), ``node_pack`` / ``edge_pack`` (fp16 tensors or None), ``id_codes`` (a Codes object over ``ids`` tagged with the edge pack, or
    None).  ``register``: enter the CSR into the layers' per-tensor cache for ``edge_index`` and tag the Codes objects with their packs, so
    that ``layer(x_codes, edge_index, identifiers=id_codes, edge_features=ef_codes)`` right behind this call launches the layer kernel
    and nothing else.  GsnError(GSN_E_UNSUPPORTED) when this launch configuration cannot write them (a graph split over workgroups)."""
    from ._index import Codes, _CSR, _cache_put
    _abi.require_gpu()
    dev = edge_index.device
    if not (edge_index.is_cuda and node_ptr.is_cuda and edge_ptr.is_cuda) or edge_index.dtype != torch.int64 or edge_index.stride(1) != 1:
        raise ValueError("count_batch_side: node_ptr, edge_ptr, edge_index int64 on the GPU (edge_index with unit column stride)")
    G = node_ptr.numel() - 1
    E = edge_index.shape[1]
    N = int(x_codes.codes.shape[0]) if x_codes is not None else (int(n_nodes) if n_nodes is not None else int(node_ptr[-1].item()))      # (one host read without either)
    rows = E if plan.mode == "edge" else N
    if out is None:
        out = torch.empty((rows, plan.n_cols), dtype=torch.int64, device=dev)
    status = torch.empty(max(G, 1), dtype=torch.int32, device=dev)
    code_status = torch.zeros(1, dtype=torch.int32, device=dev)
    side = _abi.gsn_count_side()
    csr = None
    if csr_row is not None:
        csr = _CSR()
        csr.seg_ptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        csr.perm, csr.tgt, csr.src = (torch.empty(max(E, 1), dtype=torch.int32, device=dev)[:E] for _ in range(3))
        csr._deg = csr._deg4 = None
        csr.part = None
        side.csr_row = int(csr_row); side.seg_ptr = csr.seg_ptr.data_ptr(); side.perm = csr.perm.data_ptr() if E else None
        side.sorted_target = csr.tgt.data_ptr() if E else None; side.sorted_other = csr.src.data_ptr() if E else None
        if E == 0:
            side.perm = csr.seg_ptr.data_ptr()          # (never written: no columns)
    side.n_nodes, side.n_edges = N, E
    npack = epack = None
    if x_codes is not None:
        if len(x_codes.n_classes) > 4 or sum(x_codes.n_classes) > packs.NODE_COLS - 4:
            raise ValueError("count_batch_side: <= 4 vertex code columns, <= %d classes together" % (packs.NODE_COLS - 4))
        npack = torch.empty((N, packs.NODE_COLS), dtype=torch.float16, device=dev)
        side.node_codes = x_codes.codes.data_ptr(); side.node_code_cols = len(x_codes.n_classes); side.node_clamp = int(x_codes.clamp)
        for i, c in enumerate(x_codes.n_classes):
            side.node_n_classes[i] = c
        side.node_pack = npack.data_ptr()
    enc_tab = None
    w_ids = 0
    if id_classes is not None:
        id_classes = [int(c) for c in id_classes]
        if plan.mode != "edge" or len(id_classes) != plan.n_cols:
            raise ValueError("count_batch_side: id_classes = one class count per column of an edge-mode plan")
        enc_tab = np.asarray(id_classes, dtype=np.int32)
        w_ids = sum(id_classes)
        epack = (torch.empty if ef_codes is not None else torch.zeros)((max(E, 1), packs.EDGE_COLS), dtype=torch.float16, device=dev)[:E]
    if ef_codes is not None:
        if epack is None or len(ef_codes.n_classes) > 4 or w_ids + sum(ef_codes.n_classes) > packs.EDGE_COLS or w_ids % 4 or sum(ef_codes.n_classes) > 8:
            raise ValueError("count_batch_side: edge codes ride the identifiers' edge pack (id_classes; <= %d columns together, the identifiers' a "
                             "multiple of 4, <= 8 edge code classes)" % packs.EDGE_COLS)
        side.edge_codes = ef_codes.codes.data_ptr(); side.edge_code_cols = len(ef_codes.n_classes); side.edge_clamp = int(ef_codes.clamp)
        for i, c in enumerate(ef_codes.n_classes):
            side.edge_n_classes[i] = c
        side.edge_col0 = w_ids
    side.code_status = code_status.data_ptr()
    if G > 0:
        tab = plan.device_table(dev)
        with _abi.device_guard(dev):
            rc = _abi.lib().gsn_count_encode_pack16_side_hip(_abi.ptr(plan.table), tab.data_ptr(), len(plan.table), G, node_ptr.data_ptr(), edge_ptr.data_ptr(),
                                                             edge_index.data_ptr() if E else None, edge_index.stride(0), int(bool(ids_are_global)),
                                                             int(max_nodes), int(max_edges), out.data_ptr(), status.data_ptr(), _abi.ptr(enc_tab),
                                                             int(bool(clamp)), _abi.ptr(epack) if E else None, packs.EDGE_COLS, 0, ctypes.byref(side),
                                                             _abi.current_stream())
        _abi.check(rc, "gsn_count_encode_pack16_side_hip")
    id_codes = None
    if epack is not None:
        id_codes = Codes(out, id_classes, clamp=bool(clamp), check=False)
    if register:
        if csr is not None:
            _cache_put((id(edge_index), int(csr_row), N), edge_index, csr)
        if npack is not None:
            npack._gsn_node_pack = True
            packs._claim_codes(x_codes, npack, 0)
        if id_codes is not None:
            packs._claim_codes(id_codes, epack, 0)
            if ef_codes is not None:
                packs._claim_codes(ef_codes, epack, w_ids)
    return {"ids": out, "status": status, "code_status": code_status, "csr": csr, "node_pack": npack, "edge_pack": epack, "id_codes": id_codes}


def _raise_statuses(status):
    """Read the per-graph status words back and raise what the reference raises (KeyError: utils_graph_processing.py:173)."""
    st = status.cpu().numpy()
    if (st == 1).any():
        raise KeyError("graph %d: a match maps a pattern edge onto a direction that is not a column of edge_index "
                       "(reference: utils_graph_processing.py:173)" % int(np.nonzero(st == 1)[0][0]))
    bad = np.nonzero(st > 1)[0]
    if len(bad):
        raise ValueError("graph %d: %s" % (int(bad[0]), _STATUS_MSG.get(int(st[bad[0]]), "status %d" % st[bad[0]])))


def counts2ids_batch(batch, pattern_edge_lists, mode, induced, directed_orbits=False, device=None, directed=False):
    """Batched ``subgraph_counts2ids`` over a :class:`gsn_amd.synth.Batch`-like object (node_ptr, edge_ptr, edge_index
    with batch-global ids, self loops already stripped).  -> int64 device tensor [rows_total, sum orbits]."""
    plan = CountPlan.get(pattern_edge_lists, mode, induced, directed_orbits, directed)
    out, _ = count_batch(plan, batch.node_ptr, batch.edge_ptr, batch.edge_index, ids_are_global=True, device=device)
    return out


# ------------------------------------------------------------------------------------------------------------------
# the reference's per-graph signatures
# ------------------------------------------------------------------------------------------------------------------
def _pattern_of(subgraph_dict):
    sg = subgraph_dict["subgraph"]
    if not isinstance(sg, PatternGraph):
        raise TypeError("subgraph_dict['subgraph'] must come from gsn_amd's automorphism_orbits / "
                        "induced_edge_automorphism_orbits (got %r)" % type(sg))
    return sg


def _directed_of(pats, directed):
    """The reference builds pattern and target with the same ``directed`` flag (utils_data_gen.py:38, utils_ids.py:23);
    a pattern analysed under the other flag has other orbits, so a mismatch is refused instead of counted."""
    directed = bool(directed)
    for p in pats:
        if bool(getattr(p, "directed", False)) != directed:
            raise ValueError("directed=%s, but the pattern %r was analysed with directed=%s" % (directed, p, not directed))
    return directed


def _single_graph(edge_index, num_nodes):
    ei = edge_index if isinstance(edge_index, torch.Tensor) else torch.as_tensor(np.asarray(edge_index))
    ei = ei.to(torch.int64)
    E = ei.shape[1]
    n = int(num_nodes)
    if E:
        n = max(n, int(ei.max().item()) + 1)
    return ei, n, E


def subgraph_isomorphism_vertex_counts(edge_index, **kwargs):
    """GSN-v identifiers of one graph and one pattern (utils_graph_processing.py:103-131): CPU float64 tensor
    [num_nodes, n_orbits], counts[v, o] = number of occurrences containing v at a position of orbit o.
    ``directed=True`` (:108-110): the columns of edge_index are arcs and the pattern (from
    ``automorphism_orbits(..., directed=True)``) is a digraph."""
    subgraph_dict, induced, num_nodes = kwargs["subgraph_dict"], kwargs["induced"], kwargs["num_nodes"]
    sg = _pattern_of(subgraph_dict)
    directed = _directed_of([sg], kwargs.get("directed", False))
    plan = CountPlan.get([sg.edge_list], "vertex", induced, False, directed)
    ei, n, E = _single_graph(edge_index, num_nodes)
    out, _ = count_batch(plan, [0, n], [0, E], ei, ids_are_global=False, max_nodes=n, max_edges=E)
    return out[:int(num_nodes)].cpu().to(torch.float64)


def subgraph_isomorphism_edge_counts(edge_index, **kwargs):
    """GSN-e identifiers of one graph and one pattern (utils_graph_processing.py:134-179): CPU float64 tensor
    [E, n_edge_orbits] with rows in edge_index column order."""
    subgraph_dict, induced = kwargs["subgraph_dict"], kwargs["induced"]
    if kwargs.get("directed", False):
        raise NotImplementedError("directed=True is not supported (NameError in the reference, utils_graph_processing.py:164)")
    sg = _pattern_of(subgraph_dict)
    plan = CountPlan.get([sg.edge_list], "edge", induced, sg.directed_orbits)
    ei, n, E = _single_graph(edge_index, 0)
    out, _ = count_batch(plan, [0, max(n, 1)], [0, E], ei, ids_are_global=False, max_nodes=max(n, 1), max_edges=E)
    if getattr(sg, "line_graph_orbits", False):
        _line_graph_orbits_keyerror(sg, out)
        return torch.zeros((E, len(subgraph_dict["orbit_partition"])), dtype=torch.float64)
    return out.cpu().to(torch.float64)


def _line_graph_orbits_keyerror(sg, counts):
    """Orbits from edge_automorphism_orbits (--edge_automorphism line_graph, deprecated) have ONE membership entry per
    undirected pattern edge (utils_graph_processing.py:241-243), but the reference's edge counter looks the membership up by
    position in the DIRECTED edge list (:161-173): the first match raises KeyError(m) at i = m.  Same behaviour here: a
    graph with at least one match raises, a graph without matches gets its all-zero rows."""
    if bool((counts != 0).any().item()):
        raise KeyError(len(sg.get_edges()))


def subgraph_counts2ids(count_fn, data, subgraph_dicts, subgraph_params):
    """Remove self loops, count every pattern, attach ``identifiers`` (int64) -- utils_ids.py:7-29.

    ``count_fn`` selects the mode exactly like the reference's callers do (by function identity / ``__name__``,
    utils_data_gen.py:103); all patterns are counted in one launch."""
    ei = data.edge_index
    mask = ei[0] != ei[1]
    if hasattr(data, "edge_features"):
        setattr(data, "edge_features", data.edge_features[mask])
    edge_index = ei[:, mask]
    num_nodes = data.x.shape[0]
    name = getattr(count_fn, "__name__", "")
    if name == "subgraph_isomorphism_edge_counts":
        mode = "edge"
    elif name == "subgraph_isomorphism_vertex_counts":
        mode = "vertex"
    else:
        raise TypeError("count_fn must be subgraph_isomorphism_vertex_counts or subgraph_isomorphism_edge_counts")
    pats = [_pattern_of(d) for d in subgraph_dicts]
    directed = _directed_of(pats, subgraph_params.get("directed", False))
    if directed and mode == "edge":
        raise NotImplementedError("directed=True is not supported (NameError in the reference, utils_graph_processing.py:164)")
    dirorb = any(p.directed_orbits for p in pats) if mode == "edge" else False
    plan = CountPlan.get([p.edge_list for p in pats], mode, subgraph_params["induced"], dirorb, directed)
    e_cpu, n, E = _single_graph(edge_index, num_nodes)
    out, _ = count_batch(plan, [0, n], [0, E], e_cpu, ids_are_global=False, max_nodes=n, max_edges=E)
    ids = out[:num_nodes] if mode == "vertex" else out
    if mode == "edge" and any(getattr(p, "line_graph_orbits", False) for p in pats):
        cols, blocks = 0, []
        for p, d in zip(pats, subgraph_dicts):        # per pattern, as the reference's loop would reach it
            w = CountPlan.get([p.edge_list], mode, subgraph_params["induced"], dirorb).n_cols
            if p.line_graph_orbits:
                _line_graph_orbits_keyerror(p, out[:, cols:cols + w])
                blocks.append(torch.zeros((out.shape[0], len(d["orbit_partition"])), dtype=torch.int64, device=out.device))
            else:
                blocks.append(out[:, cols:cols + w])
            cols += w
        ids = torch.cat(blocks, 1)
    setattr(data, "edge_index", edge_index)
    setattr(data, "identifiers", ids.cpu().long())
    return data

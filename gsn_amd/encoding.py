"""Encoders between the integer identifiers and the layers.

Dataset level (run once, utils_encoding.py):
    :func:`encode`, :class:`one_hot_unique`, :class:`one_hot_max` -- same call signatures and results as the reference's;
    the per-column ``np.unique(..., return_inverse=True)`` over the whole dataset runs on the device
    (gsn_column_range_hip / gsn_column_ranks_hip).

Model level (every forward, utils_graph_learning.py:44-208):
    :class:`DiscreteEmbedding` with the reference's encoder names, :class:`one_hot_encoder` (gsn_one_hot_hip),
    :class:`multi_embedding` (gsn_embed_fwd_hip / gsn_embed_bwd_hip), :class:`zero_encoder`; ``atom_encoder`` /
    ``bond_encoder`` restate ogb's AtomEncoder / BondEncoder (sum of one nn.Embedding per feature column, xavier-uniform
    initialised) with ogb's parameter names so checkpoints load.

No CPU fallback: CPU inputs are moved to the current GPU, results of the dataset-level functions come back on the CPU
because that is where the reference keeps the dataset.
"""
from __future__ import annotations

import sys

import os

import numpy as np
import torch
import torch.nn as nn

from . import _abi

# ogb.utils.features.get_atom_feature_dims() / get_bond_feature_dims() (ogb is not installed here; published constants)
ATOM_FEATURE_DIMS = [119, 4, 12, 12, 10, 6, 6, 2, 2]
BOND_FEATURE_DIMS = [5, 6, 2]
MAX_TABLE_ELEMS = 1 << 28   # presence-table entries (int32) the recoding may allocate: 1 GiB


def _device():
    _abi.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def _as_int_matrix(t):
    """Integer view of a value matrix; float inputs (the reference feeds float degrees) must hold whole numbers."""
    if t.dim() == 1:
        t = t.unsqueeze(1)
    if t.is_floating_point():
        ti = t.to(torch.int64)
        if not torch.equal(ti.to(t.dtype), t):
            raise NotImplementedError("dataset-level recoding needs integer-valued columns")
        return ti
    return t.to(torch.int64)


def column_range(values):
    """(min, max) per column of an int64 [M, C] device tensor -> two int64 device tensors [C]."""
    M, C = values.shape
    mn = torch.empty(C, dtype=torch.int64, device=values.device)
    mx = torch.empty(C, dtype=torch.int64, device=values.device)
    with _abi.device_guard(values.device):
        rc = _abi.lib().gsn_column_range_hip(M, C, values.data_ptr() if M else None, mn.data_ptr(), mx.data_ptr(),
                                             _abi.current_stream())
    _abi.check(rc, "gsn_column_range_hip")
    return mn, mx


def unique_codes(values):
    """Per-column dense ranks: ``(codes int64 [M, C], d list[int])`` with ``codes[:, c] = np.unique(values[:, c],
    return_inverse=True)[1]`` and ``d[c]`` the number of distinct values.  ``values``: integer [M, C] tensor, any device."""
    dev = _device()
    v = _as_int_matrix(values).to(dev).contiguous()
    M, C = v.shape
    if M == 0:
        return torch.zeros((0, C), dtype=torch.int64), [0] * C
    mn, mx = column_range(v)
    mn_h, mx_h = mn.cpu().numpy(), mx.cpu().numpy()          # one host read: sizes the presence tables
    sizes = (mx_h - mn_h + 1).astype(np.int64)
    base = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    if int(base[-1]) > MAX_TABLE_ELEMS:
        raise NotImplementedError("value range too wide for table recoding (%d entries > %d)" % (int(base[-1]), MAX_TABLE_ELEMS))
    base_d = torch.from_numpy(base).to(dev)
    table = torch.empty(int(base[-1]), dtype=torch.int32, device=dev)
    codes = torch.empty_like(v)
    nd = torch.empty(C, dtype=torch.int64, device=dev)
    with _abi.device_guard(dev):
        rc = _abi.lib().gsn_column_ranks_hip(M, C, v.data_ptr(), mn.data_ptr(), base_d.data_ptr(), int(base[-1]),
                                             table.data_ptr(), codes.data_ptr(), nd.data_ptr(), _abi.current_stream())
    _abi.check(rc, "gsn_column_ranks_hip")
    return codes.cpu(), [int(x) for x in nd.cpu().tolist()]


# ----------------------------------------------------------------------------------------------------------------------
# dataset level (utils_encoding.py)
# ----------------------------------------------------------------------------------------------------------------------
class one_hot_unique:
    """utils_encoding.py:37-59: ``d[c]`` = number of distinct values of column c over the whole dataset, ``fit`` hands every
    graph its rows recoded to the ranks."""

    def __init__(self, tensor_list, **kwargs):
        self._rows = [int(t.shape[0]) for t in tensor_list]
        self.codes, self.d = unique_codes(torch.cat(list(tensor_list), 0))
        self.corrs = {c: self.codes[:, c].numpy() for c in range(self.codes.shape[1])}

    def fit(self, tensor_list):
        out, ptr = [], 0
        for t in tensor_list:
            n = int(t.shape[0])
            out.append(self.codes[ptr:ptr + n].clone())
            ptr += n
        return out


class one_hot_max:
    """utils_encoding.py:62-69: ``d[c] = max + 1``; values stay as they are."""

    def __init__(self, tensor_list, **kwargs):
        cat = torch.cat(list(tensor_list), 0)
        v = _as_int_matrix(cat).to(_device()).contiguous()
        _, mx = column_range(v)
        self.d = [int(x) + 1 for x in mx.cpu().tolist()]

    def fit(self, tensor_list):
        return tensor_list


def encode(graphs, id_encoding, degree_encoding=None, **kwargs):
    """utils_encoding.py:8-34 -> ``(graphs, encoder_ids, d_id, encoder_degrees, d_degree)``; graphs are updated in place."""
    encoder_ids, d_id = None, [1] * graphs[0].identifiers.shape[1]
    if id_encoding is not None:
        fn = getattr(sys.modules[__name__], id_encoding)
        ids = [g.identifiers for g in graphs]
        encoder_ids = fn(ids, **(kwargs["ids"]))
        encoded_ids = encoder_ids.fit(ids)
        d_id = encoder_ids.d
    encoder_degrees, d_degree = None, []
    if degree_encoding is not None:
        fn = getattr(sys.modules[__name__], degree_encoding)
        degrees = [g.degrees.unsqueeze(1) for g in graphs]
        encoder_degrees = fn(degrees, **(kwargs["degree"]))
        encoded_degrees = encoder_degrees.fit(degrees)
        d_degree = encoder_degrees.d
    for i, g in enumerate(graphs):
        if id_encoding is not None:
            setattr(g, "identifiers", encoded_ids[i])
        if degree_encoding is not None:
            setattr(g, "degrees", encoded_degrees[i])
    return graphs, encoder_ids, d_id, encoder_degrees, d_degree


# ----------------------------------------------------------------------------------------------------------------------
# model level (utils_graph_learning.py)
# ----------------------------------------------------------------------------------------------------------------------
class one_hot_encoder(nn.Module):
    """utils_graph_learning.py:170-190: column c -> d_in[c] floats with a single 1."""

    def __init__(self, d_in):
        super().__init__()
        self.d_in = d_in

    def forward(self, tensor):
        from .layers import one_hot_identifiers
        return one_hot_identifiers(tensor, list(self.d_in), clamp=False)

    def __repr__(self):
        return "{}({})".format(self.__class__.__name__, self.d_in)


class zero_encoder(nn.Module):
    """utils_graph_learning.py:193-208."""

    def __init__(self, d_out):
        super().__init__()
        self.d_out = d_out

    def forward(self, tensor):
        return torch.zeros((tensor.shape[0], self.d_out), device=tensor.device)

    def __repr__(self):
        return "{}({})".format(self.__class__.__name__, self.d_out)


_META_CACHE = {}
_META_HOST = [None, 0]          # pinned staging ring [tensor, next free word]: slices are handed out once and never rewritten
_META_HOST_WORDS = 16384


def _meta_host(n):
    ring, off = _META_HOST
    if ring is None or off + n > ring.numel():
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("embedding launch inside a stream capture: the pinned staging ring for table pointers is exhausted "
                               "(run the step eagerly once before capturing it)")
        ring, off = torch.empty(_META_HOST_WORDS, dtype=torch.int64).pin_memory(), 0
    _META_HOST[0], _META_HOST[1] = ring, off + n
    return ring[off:off + n]


def _table_meta(tables, device):
    """Device array [table pointers | row counts] of an embedding launch.  It travels through a slice of a pinned staging ring with an
    asynchronous copy (from pageable memory the host would wait for the stream, and the copy could not be recorded into a graph
    capture); the arrays are kept per pointer tuple -- parameters keep their addresses, and the caching allocator hands a step's
    gradient tables the addresses of the step before.  Inside a capture (gsn_amd.graphs) the copy becomes a node of the graph that
    re-reads the pinned slice at every replay: slices are never rewritten, entries a capture has used are never evicted, and an
    entry MADE inside a capture (its device array is only written by replays) is keyed on that capture."""
    from . import _abi
    ptrs = tuple([t.data_ptr() for t in tables] + [int(t.shape[0]) for t in tables])
    cap = 0
    if torch._C._cuda_isCurrentStreamCapturing():
        with _abi.device_guard(device):
            cap = int(_abi.lib().gsn_stream_capture_id(_abi.current_stream()))
    hit = _META_CACHE.get((ptrs, str(device), 0))
    if hit is None and cap:
        hit = _META_CACHE.get((ptrs, str(device), cap))
    if hit is None:
        if len(_META_CACHE) > 1024:
            for k in [k for k, v in _META_CACHE.items() if not v[2]]:
                del _META_CACHE[k]
        host = _meta_host(len(ptrs))
        host.copy_(torch.tensor(ptrs, dtype=torch.int64))
        dev_t = torch.empty(len(ptrs), dtype=torch.int64, device=device)
        dev_t.copy_(host, non_blocking=True)
        hit = _META_CACHE[(ptrs, str(device), cap)] = [dev_t, host, bool(cap)]
    elif cap:
        hit[2] = True           # a graph holds its address now
    return hit[0]


# Out-of-range codes.  The reference's nn.Embedding raises IndexError (CPU) / a device-side assert that surfaces later (GPU).  Here the
# kernel raises a status word and makes the row NaN; reading the word back at once would synchronise host and device at EVERY
# embedding (twelve times per step of the ogb model).  So the word travels to pinned host memory behind the kernel and is looked at when
# its event has completed -- at the next embedding call, in the backward, or by check_embedding_status() -- i.e. the IndexError arrives
# one call late, like any asynchronous device error.  GSN_EMBED_STATUS_SYNC=1 checks at once.
EMBED_STATUS_SYNC = os.environ.get("GSN_EMBED_STATUS_SYNC", "0") == "1"
EMBED_BWD_FLAT = os.environ.get("GSN_EMBED_BWD_FLAT", "1") != "0"      # (0: gradient tables through a device pointer array, A/B)
_PENDING_STATUS = []
_STATUS_RING, _STATUS_NEXT = None, 0


def check_embedding_status(wait=False):
    """Raise IndexError if an embedding launch whose result has arrived saw a code outside its table (wait=True: of any launch so far)."""
    if _PENDING_STATUS and torch.cuda.is_current_stream_capturing():
        return                              # (no event queries while a HIP graph is being captured)
    while _PENDING_STATUS and (wait or _PENDING_STATUS[0][0].query()):
        ev, host = _PENDING_STATUS.pop(0)
        if wait:
            ev.synchronize()
        if int(host[0]) != 0:
            _PENDING_STATUS.clear()
            raise IndexError("index out of range in embedding table")


def _drain_at_exit():
    # the status of the LAST embedding launches of a process (a final eval forward) is otherwise never looked at
    try:
        check_embedding_status(wait=True)
    except IndexError as e:
        import sys
        print("gsn_amd.encoding: %s (reported at interpreter exit: the launch was asynchronous)" % e, file=sys.stderr)
    except Exception:
        pass


import atexit  # noqa: E402
atexit.register(_drain_at_exit)


def _defer_status(status):
    if EMBED_STATUS_SYNC:
        if int(status.item()) != 0:
            raise IndexError("index out of range in embedding table")
        return
    global _STATUS_RING, _STATUS_NEXT
    if torch.cuda.is_current_stream_capturing():
        return                              # (a captured step cannot report: its out-of-range rows are NaN, gsn_amd.graphs)
    if _STATUS_RING is None:
        _STATUS_RING = torch.zeros(128, dtype=torch.int32).pin_memory()      # (one pinned allocation: slots handed out in turn)
    if len(_PENDING_STATUS) >= 96:
        check_embedding_status(wait=True)
    host = _STATUS_RING[_STATUS_NEXT:_STATUS_NEXT + 1]
    _STATUS_NEXT = (_STATUS_NEXT + 1) % 128
    host.copy_(status, non_blocking=True)
    # (the event goes onto the stream the copy was issued on -- the current stream of the status word's device, which need not be the
    #  current device: an event of another device's stream could complete before the copy lands and a stale 0 would be read)
    with torch.cuda.device(status.device):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(status.device))
    _PENDING_STATUS.append((ev, host))


class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, codes, concat, *tables):
        check_embedding_status()
        dev = codes.device
        d = int(tables[0].shape[1])
        M, C = codes.shape
        tabs = [t if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().to(torch.float32).contiguous() for t in tables]
        meta = _table_meta(tabs, dev)
        out = torch.empty((M, C * d if concat else d), dtype=torch.float32, device=dev)
        from .layers import _zeros
        status = _zeros(1, torch.int32, dev)
        rows = np.ascontiguousarray([t.shape[0] for t in tabs], dtype=np.int64)
        from ._runtime import _timed
        with _abi.device_guard(dev), _timed("embed_fwd", 8.0 * M * C + 4.0 * out.numel()):
            rc = _abi.lib().gsn_embed_fwd_hip(M, C, d, int(concat), codes.data_ptr() if M else None, meta.data_ptr(),
                                              _abi.ptr(rows), out.data_ptr() if M else None, status.data_ptr(),
                                              _abi.current_stream())
        _abi.check(rc, "gsn_embed_fwd_hip")
        _defer_status(status)
        ctx.save_for_backward(codes)
        ctx.concat, ctx.shapes = concat, [tuple(t.shape) for t in tables]
        return out

    @staticmethod
    def backward(ctx, gout):
        check_embedding_status()
        (codes,) = ctx.saved_tensors
        dev = codes.device
        M, C = codes.shape
        d = ctx.shapes[0][1]
        # gradient tables: ONE zero fill for all of them (nine tables of the atom encoder: nine launches otherwise); offsets rounded to
        # 16 bytes so that every table keeps the alignment of a tensor of its own
        sizes = [(s[0] * s[1] + 3) // 4 * 4 for s in ctx.shapes]
        rows = np.ascontiguousarray([s[0] for s in ctx.shapes], dtype=np.int64)
        g = gout.to(torch.float32).contiguous()
        with _abi.device_guard(dev):
            flat_ok = EMBED_BWD_FLAT and bool(_abi.lib().gsn_embed_bwd_flat_supported(max(M, 1), C, int(ctx.concat), _abi.ptr(rows)))
        if flat_ok:
            # the tables' addresses as launch arguments (base + host offsets): no device pointer array -- no copy per call, no memcpy node per
            # embedding in a captured step -- and therefore nothing that needs the tables at the addresses of the step before: zeros from the arena
            from .layers import _zeros
            flat = _zeros(sum(sizes), torch.float32, dev)
        else:
            # (its own allocation, not the zero arena: the caching allocator hands a step the addresses of the step before, which keeps the
            #  pointer arrays of _table_meta cached -- arena slices move every step)
            flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        grads, offs, o = [], [], 0
        for s_, n_ in zip(ctx.shapes, sizes):
            grads.append(flat[o:o + s_[0] * s_[1]].view(s_))
            offs.append(o)
            o += n_
        if flat_ok:
            offs = np.ascontiguousarray(offs, dtype=np.int64)
            with _abi.device_guard(dev):
                rc = _abi.lib().gsn_embed_bwd_flat_hip(M, C, d, int(ctx.concat), codes.data_ptr() if M else None, flat.data_ptr(),
                                                       _abi.ptr(offs), _abi.ptr(rows), g.data_ptr() if M else None, _abi.current_stream())
            _abi.check(rc, "gsn_embed_bwd_flat_hip")
            return (None, None) + tuple(grads)
        meta = _table_meta(grads, dev)
        with _abi.device_guard(dev):
            rc = _abi.lib().gsn_embed_bwd_hip(M, C, d, int(ctx.concat), codes.data_ptr() if M else None, meta.data_ptr(),
                                              _abi.ptr(rows), g.data_ptr() if M else None, _abi.current_stream())
        _abi.check(rc, "gsn_embed_bwd_hip")
        return (None, None) + tuple(grads)


def embed_columns(codes, tables, concat):
    """``concat_c tables[c][codes[:, c]]`` or the sum over c, on the HIP kernel; differentiable w.r.t. the tables."""
    _abi.require_gpu()
    if not codes.is_cuda:
        raise RuntimeError("embed_columns: codes must live on the GPU (no CPU fallback)")
    codes = codes.to(torch.int64).contiguous()
    if codes.dim() == 1:
        codes = codes.unsqueeze(1)
    if codes.shape[1] != len(tables):
        raise ValueError("need one table per code column")
    return _EmbedFn.apply(codes, bool(concat), *tables)


class multi_embedding(nn.Module):
    """utils_graph_learning.py:134-167: one ``nn.Embedding(d_in[i], d_out)`` per column (parameters ``encoder.{i}.weight``),
    concatenated or summed."""

    def __init__(self, d_in, d_out, aggr="concat", init=None):
        super().__init__()
        self.d_in = d_in
        self.aggr = aggr
        enc = []
        for i in range(len(d_in)):
            enc.append(nn.Embedding(d_in[i], d_out))
            if init == "zeros":
                nn.init.constant_(enc[i].weight.data, 0)
            else:
                nn.init.xavier_uniform_(enc[-1].weight.data)
        self.encoder = nn.ModuleList(enc)

    def forward(self, tensor):
        if self.aggr not in ("concat", "sum"):
            raise NotImplementedError("multi embedding aggregation {} is not currently supported.".format(self.aggr))
        n = tensor.shape[1]
        return embed_columns(tensor, [self.encoder[i].weight for i in range(n)], self.aggr == "concat")


class _FeatureSumEncoder(nn.Module):
    """ogb.graphproppred.mol_encoder.AtomEncoder / BondEncoder: sum over feature columns of an xavier-initialised
    nn.Embedding each."""

    def __init__(self, emb_dim, dims, list_name):
        super().__init__()
        embs = nn.ModuleList()
        for dim in dims:
            e = nn.Embedding(dim, emb_dim)
            nn.init.xavier_uniform_(e.weight.data)
            embs.append(e)
        setattr(self, list_name, embs)
        self._list_name = list_name

    def forward(self, x):
        embs = getattr(self, self._list_name)
        return embed_columns(x, [embs[i].weight for i in range(x.shape[1])], False)


class AtomEncoder(_FeatureSumEncoder):
    def __init__(self, emb_dim):
        super().__init__(emb_dim, ATOM_FEATURE_DIMS, "atom_embedding_list")


class BondEncoder(_FeatureSumEncoder):
    def __init__(self, emb_dim):
        super().__init__(emb_dim, BOND_FEATURE_DIMS, "bond_embedding_list")


class DiscreteEmbedding(nn.Module):
    """utils_graph_learning.py:44-130, same encoder names, ``d_out`` and parameter names."""

    def __init__(self, encoder_name, d_in_features, d_in_encoder, d_out_encoder, **kwargs):
        super().__init__()
        from . import layers
        kwargs["init"] = None if "init" not in kwargs else kwargs["init"]
        self.encoder_name = encoder_name
        if encoder_name == "zero_encoder":
            self.encoder = zero_encoder(d_out_encoder)
            d_out = d_out_encoder
        elif encoder_name == "linear":
            self.encoder = nn.Linear(d_in_features, d_out_encoder, bias=True)
            d_out = d_out_encoder
        elif encoder_name == "mlp":
            self.encoder = layers.mlp(d_in_features, d_out_encoder, d_out_encoder, kwargs["seed"], kwargs["activation_mlp"],
                                      kwargs["bn_mlp"])
            d_out = d_out_encoder
        elif encoder_name == "one_hot_encoder":
            self.encoder = one_hot_encoder(d_in_encoder)
            d_out = sum(d_in_encoder)
        elif encoder_name == "embedding":
            self.encoder = multi_embedding(d_in_encoder, d_out_encoder, kwargs["aggr"], kwargs["init"])
            d_out = len(d_in_encoder) * d_out_encoder if kwargs["aggr"] == "concat" else d_out_encoder
        elif encoder_name == "atom_one_hot_encoder":
            dims = ATOM_FEATURE_DIMS if kwargs["features_scope"] == "full" else ATOM_FEATURE_DIMS[:2]
            self.encoder = one_hot_encoder(dims)
            d_out = sum(dims)
        elif encoder_name == "bond_one_hot_encoder":
            dims = BOND_FEATURE_DIMS if kwargs["features_scope"] == "full" else BOND_FEATURE_DIMS[:2]
            self.encoder = one_hot_encoder(dims)
            d_out = sum(dims)
        elif encoder_name == "atom_encoder":
            self.encoder = AtomEncoder(d_out_encoder)
            d_out = d_out_encoder
        elif encoder_name == "bond_encoder":
            self.encoder = BondEncoder(emb_dim=d_out_encoder)
            d_out = d_out_encoder
        elif encoder_name == "None":
            self.encoder = None
            d_out = d_in_features
        else:
            raise NotImplementedError("Encoder {} is not currently supported.".format(encoder_name))
        self.d_out = d_out

    def forward(self, x):
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        if self.encoder is None:
            return x.float()
        if self.encoder_name == "linear":
            from .layers import run_linear_module
            return run_linear_module(self.encoder, x.float())
        if self.encoder_name == "mlp":
            return self.encoder(x.float())
        return self.encoder(x.long())

"""Exact fp16 row packs: the inputs of a model's first GSN layer a second time, in the layout the matrix pipe reads.

Layer 0 of every reference model reads one-hot / small-integer encodings (``DiscreteEmbedding('one_hot_encoder')``,
utils_graph_learning.py:78-88 / :170-187; the counts' one-hot rows): every value is exact in fp16.  A producer that knows this
writes its rows twice -- the fp32 tensor the reference's signatures carry, and the same values as fp16 into a *pack*:

* node pack  fp16 [N, 32]: columns 0 .. d_x-1 = x, d_x .. 30 = 0, column 31 = 1.0 (it carries the edge stage's bias)
* edge pack  fp16 [E, 16]: the per-edge tensors (identifiers, edge features) side by side, zero padded

and TAGS the fp32 tensor with it (``tensor._gsn_pack16``).  The one-launch ``general`` layer (csrc/layer_rp.hip, through
``gsn_layer_fused_fwd_pack16_hip``) then gathers five 16-byte fragments per edge row instead of ten fp32 quads that it has to
convert and test for exactness.  Nothing changes for tensors without a tag, for a tensor modified since (its ``_version`` moved), or
for a pack whose columns another tensor has claimed since: those take the fp32 kernel (csrc/layer_rr.hip).

Producers: :func:`gsn_amd.counting.count_batch` (``encoded_pack=``: the counting kernel writes the fp16 rows itself),
:func:`node_pack` / :func:`edge_pack` over existing fp32 tensors (``gsn_pack16_rows_hip``: converts and CHECKS -- a value that is
not exact in fp16 or not below 2 in magnitude raises ``ValueError`` and nothing is tagged).
"""
from __future__ import annotations

import weakref

import torch

from . import _abi
from ._runtime import INPUT_EPOCH

NODE_COLS = 32      # fp16 columns of a node pack row (64 bytes)
EDGE_COLS = 16      # fp16 columns of an edge pack row (32 bytes)


def new_node_pack(n_rows, device):
    return torch.zeros((n_rows, NODE_COLS), dtype=torch.float16, device=device)


def new_edge_pack(n_rows, device):
    return torch.zeros((n_rows, EDGE_COLS), dtype=torch.float16, device=device)


def claim(t, pack, col0):
    """Tag ``t`` as the fp32 original of columns col0 .. col0 + t.shape[1] of ``pack`` (called by a producer right after it wrote both)."""
    owners = getattr(pack, "_gsn_owners", None)
    if owners is None:
        owners = {}
        pack._gsn_owners = owners
    w = t.shape[1] if t.dim() == 2 else 1
    for c in [c for c, (ref, cw) in owners.items() if c < col0 + w and col0 < c + cw]:      # overlapping earlier claims end here
        del owners[c]
    owners[int(col0)] = (weakref.ref(t), w)
    t._gsn_pack16 = (pack, int(col0), t._version, INPUT_EPOCH[0])


def _claim_codes(cd, pack, col0):
    """The same for a gsn_amd.layers.Codes object whose one-hot encoding was just written into columns col0.. of ``pack``: the claim
    enters the pack's owner table (a later producer that writes those columns ends it), the tag is bound to the code tensor's version."""
    owners = getattr(pack, "_gsn_owners", None)
    if owners is None:
        owners = {}
        pack._gsn_owners = owners
    w = sum(cd.n_classes)
    for c in [c for c, (ref, cw) in owners.items() if c < col0 + w and col0 < c + cw]:
        del owners[c]
    owners[int(col0)] = (weakref.ref(cd), w)
    cd._pack16 = (pack, int(col0), cd.codes._version, INPUT_EPOCH[0])


def release(t):
    """Drop the tag of ``t`` (a kernel is about to overwrite it through its raw pointer: the version counter will not move)."""
    if getattr(t, "_gsn_pack16", None) is not None:
        t._gsn_pack16 = None


def tag_of(t, n_rows, n_cols):
    """(pack, col0) when ``t`` carries a current tag for a pack of ``n_cols`` columns, else None."""
    tag = getattr(t, "_gsn_pack16", None)
    if tag is None:
        return None
    pack, col0, ver, epoch = tag
    if epoch != INPUT_EPOCH[0] or ver != t._version or pack.device != t.device or pack.shape != (n_rows, n_cols) or pack.dtype != torch.float16 or not pack.is_contiguous():
        return None
    own = getattr(pack, "_gsn_owners", {}).get(col0)
    if own is None or own[0]() is not t:
        return None
    return pack, col0


def _pack_rows(t, pack, col0, one_col, check):
    src = t.detach()
    if src.dtype != torch.float32 or not src.is_contiguous():
        src = src.float().contiguous()
    if src.dim() == 1:
        src = src.unsqueeze(-1)
    status = torch.zeros(1, dtype=torch.int32, device=t.device)
    with _abi.device_guard(t.device):
        _abi.check(_abi.lib().gsn_pack16_rows_hip(src.data_ptr(), src.shape[0], src.shape[1], pack.data_ptr(), pack.shape[1], col0, one_col,
                                                  status.data_ptr(), _abi.current_stream()), "gsn_pack16_rows_hip")
    if check and int(status.item()) != 0:
        raise ValueError("rows are not exact in fp16 or not below 2 in magnitude: no fp16 pack for this tensor")


def node_pack(x, check=True):
    """fp16 node pack of ``x`` [N, d_x <= 28]; tags ``x``.  ``check=False`` skips the read-back of the exactness flag (the caller
    vouches for the rows, e.g. one-hot rows it has just made)."""
    if x.dim() != 2 or x.shape[1] > NODE_COLS - 4:
        raise ValueError("node_pack: x must be [N, d_x] with d_x <= %d" % (NODE_COLS - 4))
    pack = new_node_pack(x.shape[0], x.device)
    _pack_rows(x, pack, 0, NODE_COLS - 1, check)
    pack._gsn_node_pack = True
    claim(x, pack, 0)
    return pack


def edge_pack(tensors, check=True, pack=None):
    """ONE fp16 edge pack for the per-edge tensors of a layer call, in the order the layer concatenates them (identifiers, then edge
    features: GSN_edge_sparse.py:160-165); tags each of them.  ``pack``: write into an existing pack (its other columns are kept)."""
    tensors = [t for t in tensors if t is not None]
    rows = tensors[0].shape[0]
    if sum(t.shape[1] for t in tensors) > EDGE_COLS:
        raise ValueError("edge_pack: more than %d columns" % EDGE_COLS)
    if pack is None:
        pack = new_edge_pack(rows, tensors[0].device)
    col = 0
    for t in tensors:
        _pack_rows(t, pack, col, -1, check)
        col += t.shape[1]
    col = 0
    for t in tensors:
        claim(t, pack, col)
        col += t.shape[1]
    return pack


def _pack_codes(cd, pack, col0, one_col, check=None):
    """``check`` (default: gsn_amd.flags.CODE_STATUS_CHECK): read the out-of-range flag back -- IndexError, as the weight-row-gather stage
    and the reference's F.one_hot; clamped codes cannot be out of range and are never read back."""
    import numpy as np
    from . import flags, layers
    ncls = np.ascontiguousarray(cd.n_classes, dtype=np.int32)
    if check is None:
        check = flags.CODE_STATUS_CHECK if cd.check is None else cd.check
    check = check and not cd.clamp
    if check and torch.cuda.is_current_stream_capturing():
        check = False          # (nothing can be read back inside a stream capture; an out-of-range code leaves a zero block, as the dense encoder)
    status = torch.zeros(1, dtype=torch.int32, device=cd.codes.device) if check else None
    with _abi.device_guard(cd.codes.device):
        _abi.check(_abi.lib().gsn_one_hot_pack16_hip(cd.codes.shape[0], cd.codes.shape[1], cd.codes.data_ptr(), _abi.ptr(ncls), int(cd.clamp),
                                                    pack.data_ptr(), pack.shape[1], int(col0), int(one_col), _abi.ptr(status), _abi.current_stream()),
                   "gsn_one_hot_pack16_hip")
    if check and int(status.item()) != 0:
        raise IndexError("a code outside its column's classes (one-hot encoding of integer codes)")


def pack_node_codes(cd, pack=None):
    """Node pack of the one-hot encoding of ``cd`` (gsn_amd.layers.Codes); kept on the Codes object (codes are immutable by convention).
    ``pack``: reuse a pack made by new_node_pack for codes of the same width (its zero columns are not rewritten)."""
    if sum(cd.n_classes) > NODE_COLS - 4:
        raise ValueError("pack_node_codes: more than %d encoded columns" % (NODE_COLS - 4))
    if pack is None:
        pack = new_node_pack(cd.codes.shape[0], cd.codes.device)
    elif tuple(pack.shape) != (cd.codes.shape[0], NODE_COLS) or pack.dtype != torch.float16:
        raise ValueError("pack_node_codes: pack must be fp16 [%d, %d]" % (cd.codes.shape[0], NODE_COLS))
    _pack_codes(cd, pack, 0, NODE_COLS - 1)
    pack._gsn_node_pack = True          # (column 31 = 1.0 has been written: the edge stage's bias rides it)
    _claim_codes(cd, pack, 0)
    return pack


def pack_edge_codes(cd, pack, col0):
    """The one-hot encoding of ``cd`` into columns col0.. of the edge pack ``pack`` (its other columns are kept); kept on the Codes object."""
    if col0 + sum(cd.n_classes) > EDGE_COLS or pack.shape != (cd.codes.shape[0], EDGE_COLS):
        raise ValueError("pack_edge_codes: columns %d .. %d of a %s pack" % (col0, col0 + sum(cd.n_classes), tuple(pack.shape)))
    _pack_codes(cd, pack, col0, -1)
    _claim_codes(cd, pack, col0)
    return pack


def from_codes(x_codes, per_edge):
    """(node pack, edge pack or None) of a layer call whose inputs are integer codes (gsn_amd.layers.Codes: the one-hot encoder's INPUT) --
    the dense fp32 one-hot rows are never written.  ``per_edge``: the edge-level inputs in concatenation order, each a Codes or an fp32
    tensor that carries a current pack tag (the counting kernel's encoded identifiers).  Codes already encoded (pack_node_codes /
    pack_edge_codes) are not encoded again.  None when the widths or the tags do not fit together."""
    per_edge = [c for c in per_edge if c is not None]
    if sum(x_codes.n_classes) > NODE_COLS - 4 or sum(_width(c) for c in per_edge) > EDGE_COLS:
        return None
    tgx = _codes_tag(x_codes)
    npk = tgx[0] if tgx is not None and tgx[1] == 0 and getattr(tgx[0], "_gsn_node_pack", False) else pack_node_codes(x_codes)
    epk = None
    if per_edge:
        rows = per_edge[0].shape[0]
        col, todo = 0, []
        for c in per_edge:      # the pack the already-encoded inputs live in (all the same one, columns in order)
            tg = tag_of(c, rows, EDGE_COLS) if isinstance(c, torch.Tensor) else _codes_tag(c)
            if tg is None:
                if isinstance(c, torch.Tensor):
                    return None
                todo.append((c, col))
            elif tg[1] != col or (epk is not None and tg[0] is not epk) or tuple(tg[0].shape) != (rows, EDGE_COLS):
                return None
            else:
                epk = tg[0]
            col += _width(c)
        if epk is None:
            epk = new_edge_pack(rows, x_codes.codes.device)
        for c, col in todo:
            pack_edge_codes(c, epk, col)
    return npk, epk


def _codes_tag(cd):
    """(pack, first column, ..) of an encoded Codes object while its code tensor is unchanged (an in-place write moves the version counter)"""
    tg = cd._pack16
    if tg is None or tg[2] != cd.codes._version or tg[3] != INPUT_EPOCH[0]:      # (the epoch: a tag made before a stream capture began does not count inside it)
        return None
    own = getattr(tg[0], "_gsn_owners", {}).get(tg[1])      # (another Codes / tensor encoded into these columns since: the claim is gone)
    return tg if own is not None and own[0]() is cd else None


def _width(c):
    return c.shape[1] if isinstance(c, torch.Tensor) else sum(c.n_classes)


def lookup(x, per_edge):
    """(node pack, edge pack or None) when ``x`` and every tensor of ``per_edge`` (in concatenation order) carry current tags that
    fit together -- one edge pack, columns in order from 0 --, else None."""
    if not isinstance(x, torch.Tensor) or x.dim() != 2:
        return None
    nt = tag_of(x, x.shape[0], NODE_COLS)
    if nt is None or nt[1] != 0 or not getattr(nt[0], "_gsn_node_pack", False):      # (a node pack: made by node_pack / pack_node_codes, column 31 = 1.0)
        return None
    epack, col = None, 0
    for t in per_edge:
        if not isinstance(t, torch.Tensor) or t.dim() != 2:
            return None
        et = tag_of(t, t.shape[0], EDGE_COLS)
        if et is None or et[1] != col or (epack is not None and et[0] is not epack):
            return None
        epack = et[0]
        col += t.shape[1]
    return nt[0], epack

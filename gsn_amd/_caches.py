"""Derived-weight caches of the layers (folded weights, fp16 planes, prepared fragment buffers, eval-mode BatchNorm vectors) and what keeps
them honest: version-keyed entries, the synchronous and the asynchronous fingerprint validation, the bookkeeping around stream captures."""
from __future__ import annotations

import time
import warnings
import weakref

import torch

from . import _abi, flags
from ._index import _CSR_CACHE
from ._runtime import _zeros

def _module_fingerprint(module):
    """Validation mode (``GSN_VALIDATE_CACHES=1`` / ``flags.VALIDATE_CACHES = True``): a content fingerprint of every floating-point
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
        if now - st["t_last"] < flags.ASYNC_VALIDATE_INTERVAL and (not pend or not pend[0][0].query()):
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
        if now - st["t_last"] < flags.ASYNC_VALIDATE_INTERVAL:
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
    # the tags that live ON input objects (a Codes object's fp16 pack and dense rows, an fp32 tensor's pack tag) cannot be enumerated: they carry
    # the epoch they were made in and are rejected once it has moved
    from ._runtime import INPUT_EPOCH
    INPUT_EPOCH[0] += 1


def invalidate_caches(module=None):
    """Drop the derived tensors this module keeps per parameter VERSION (folded first weight of the `general` layers,
    eval-mode BatchNorm scale / shift vectors, transposed weights) -- needed only after writing a parameter or buffer
    through ``.data`` (``p.data.copy_`` / ``fill_``), which PyTorch does not count as a new version; optimizers,
    ``load_state_dict`` and ordinary in-place ops do bump the version and need no call.  ``module=None``: also the CSR cache."""
    if module is None:
        _CSR_CACHE.clear()
        return
    for m in module.modules():
        for attr in ("_fold_cache", "_split_cache", "_fused_prep", "_fused_prep16", "_fplan", "_gsn_eval_cache", "_gsn_wt"):
            if hasattr(m, attr):
                try:
                    delattr(m, attr)
                except AttributeError:
                    pass
        for prm in m.parameters(recurse=False):          # fp16 planes of the weights (gsn_linear_f16x3_prepare_hip)
            if hasattr(prm, "_gsn_f16x3"):
                del prm._gsn_f16x3

"""Small run-time helpers shared by the host modules: activation codes, the per-family HIP-event timers (bench.py's hook), the zero
arenas, contiguity / device checks."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _abi, flags

_ACT_CODE = {"identity": 0, "relu": 1, "elu": 2, "tanh": 3}
_MAX_BLOCKS = 5


class _timed:
    def __init__(self, name, work=0.0):
        self.name, self.work = name, work
        self.on = flags.KERNEL_TIMER is not None and (flags.KERNEL_TIMER_ONLY is None or name in flags.KERNEL_TIMER_ONLY)

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            flags.KERNEL_TIMER.setdefault(self.name, []).append((self.e0, self.e1, self.work))
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
_ZARENA_TIERS = ((256 * 1024, 64 * 1024), (int(os.environ.get("GSN_ZERO_ARENA_MB", "32")) * 1024 * 1024, 2 * 1024 * 1024))     # (arena bytes, largest request served from it; the large tier: 8 MiB until the split-K outputs of a small-batch backward -- 18 x 1-2 MB per step -- made it four fills per step)


_ITEMSIZE = {torch.float64: 8, torch.float32: 4, torch.int64: 8, torch.int32: 4, torch.float16: 2, torch.uint8: 1}


def _zeros(n, dtype, device):
    """1-D zero tensor of ``n`` elements of ``dtype`` on ``device`` (cuda) from the arenas: small requests (statistics, status words) from a
    256 KiB arena, the weight-gradient accumulators of a dense backward (up to 2 MiB) from an 8 MiB one -- a d = 300 training step asks for
    ~20 of those, one fill of 8 MiB costs what one fill of 700 KiB does."""
    item = _ITEMSIZE[dtype]
    nbytes = (n * item + 255) // 256 * 256
    tier = 0 if nbytes <= 65536 else (1 if nbytes <= 2097152 else -1)
    if tier < 0 or device.type != "cuda" or not flags.ZERO_ARENA:
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


# Epoch of everything cached per INPUT tensor's contents (the CSR of an edge_index, the fp16 pack / dense one-hot rows of a Codes object, the
# pack tag of an fp32 tensor).  gsn_amd.graphs bumps it in front of a stream capture (drop_input_caches): a tag made before -- by a warm-up step
# on the same static input objects -- is then stale BY EPOCH, so the capture records the producer launch (encoder, pack, index build) instead of
# finding a valid tag and recording only the consumer, which would replay on the warm-up's rows after ``static.copy_(new_batch)``.
INPUT_EPOCH = [0]

"""HIP-graph replay of a fixed-shape step (the reference's real batch sizes are 32-128 graphs).

At those sizes a step is ~10 small dependent launches and the time goes to enqueueing them from Python (bench.py `small_batch`: 120 us eager
at B = 128).  Every launch of this package goes to PyTorch's current stream, so a step can be captured once and replayed (66-72 us):

    step = gsn_amd.graphs.GraphedStep(lambda: model_forward(static_x, static_ei, ...))   # warm-up on a side stream, then capture
    static_x.copy_(new_x); ...                                                          # refill the SAME input tensors
    y = step()                                                                          # one hipGraphLaunch; y is the captured output

Rules of stream capture apply: shapes, pointers and launch arguments are frozen (re-capture for another batch shape), nothing inside the step
may synchronise (status read-backs are deferred or off: `check=False` for the CSR build, the embedding status ring), caches that the step
would FILL must be filled before the capture (the warm-up runs do that: prepared weights, CSR of a registered partition, constant rows).
"""
from __future__ import annotations

import torch


class GraphedStep:
    """Capture ``fn()`` (no arguments: it closes over its static input tensors) into a HIP graph; calling the object replays it and
    returns the outputs of the captured run (the same tensor objects every time -- copy them out before the next replay if needed)."""

    def __init__(self, fn, warmup: int = 3, device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()
        self.replays = 0

    def __call__(self):
        self.graph.replay()
        self.replays += 1
        return self.outputs

    replay = __call__

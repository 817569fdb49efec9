"""HIP-graph replay of a fixed-shape step (the reference's real batch sizes are 32-128 graphs).

At those sizes a step is ~10 small dependent launches and the time goes to enqueueing them from Python (bench.py `small_batch`: 120 us eager
at B = 128).  Every launch of this package goes to PyTorch's current stream, so a step can be captured once and replayed (66-72 us):

    step = gsn_amd.graphs.GraphedStep(lambda: model_forward(static_x, static_ei, ...))   # warm-up on a side stream, then capture
    static_x.copy_(new_x); ...                                                          # refill the SAME input tensors
    y = step()                                                                          # one hipGraphLaunch; y is the captured output

Rules of stream capture apply: shapes, pointers and launch arguments are frozen (re-capture for another batch shape), nothing inside the step
may synchronise (status read-backs are deferred or off: `check=False` for the CSR build, the embedding status ring).  Caches keyed on
PARAMETERS (prepared weights, constant rows) are filled by the warm-up runs and stay valid; caches keyed on the INPUT tensors (the
aggregation index of `edge_index`, readout index pairs, graph sizes) are dropped right before the capture, so that the kernels that build
them become nodes of the graph: a replay after `static_edge_index.copy_(new_batch)` rebuilds them from the new contents.
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
        from . import flags, layers
        layers.drop_input_caches()                     # (the index builds of the static inputs are recorded, not looked up)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()
        layers.drop_input_caches()                     # (what the capture cached lives in graph-pool memory no eager call has written:
        layers.drop_capture_caches()                   #  indices of the inputs, and the derived weights keyed on the parameters' versions)
        self.replays = 0

    def __call__(self):
        self.graph.replay()
        self.replays += 1
        return self.outputs

    replay = __call__


class GraphedTrainStep:
    """One whole training step -- ``optimizer.zero_grad(); loss = loss_fn(); loss.backward(); all-reduce; optimizer.step()``
    (train_test_funcs.py:88-106) -- captured into ONE HIP graph and replayed with one launch.

    At the reference's batch sizes (README.md:112, :121: 128 / 32 graphs) a step is several hundred launches of a few microseconds
    each; eager, its time is the Python in front of them (profiles/r03_train_step_progress.txt: 6.6 ms at B = 32 of which < 1 ms is
    device time).  Captured, forward, the native adjoints, the gradient bucket (+ the RCCL all-reduce when the process group has more
    than one rank) and the optimizer's multi-tensor update are nodes of one graph.

        step = GraphedTrainStep(lambda: loss_fn(model(data), data.y), optimizer)     # `data`: static tensors, refilled with copy_()
        loss = step()                                                                # replay; `loss` is the captured 0-d tensor

    What stays correct across replays, and why:
      * BatchNorm running statistics and ``num_batches_tracked`` live on the device and are advanced by the captured kernels;
      * dropout draws from PyTorch's generator, whose Philox offset the graph advances per replay (a replay is not a repeat);
      * the parameters are updated in place by the captured optimizer kernels, which PyTorch's version counters do not see: the
        derived-weight caches of gsn_amd.layers are keyed on those counters, so every replay bumps them -- parameters and the BatchNorm
        buffers the captured kernels write (torch.autograd.graph.increment_version -- no launch) -- and an eager forward after a replay
        prepares its weights and eval-mode BatchNorm vectors afresh.
    The warm-up steps are REAL steps (they update the parameters); shapes are frozen: capture one object per batch shape.
    Optimizers with a host-side step counter (Adam, AdamW ...) must be built with ``capturable=True``."""

    def __init__(self, loss_fn, optimizer, parameters=None, warmup: int = 3, allreduce: bool = True, average: bool = True, group=None,
                 force_allreduce: bool = False, device=None):
        from . import dist as gdist
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if optimizer.defaults.get("capturable") is False:
            raise ValueError("GraphedTrainStep: build %s with capturable=True (its step counter must live on the device)" % type(optimizer).__name__)
        self.optimizer = optimizer
        self.params = [p for g in optimizer.param_groups for p in g["params"]] if parameters is None else list(parameters)

        def step():
            optimizer.zero_grad(set_to_none=True)
            loss = loss_fn()
            loss.backward()
            if allreduce:
                gdist.allreduce_gradients(self.params, average=average, group=group, force=force_allreduce)
            optimizer.step()
            return loss.detach()

        self._eager = step
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)          # (the captured backward allocates the gradients from the graph's pool)
        from . import flags, layers
        flags.RAW_WRITTEN = []
        # caches keyed on the INPUT tensors (aggregation index, readout pairs, graph sizes) were filled by the warm-up steps: dropped, so
        # that their builds are recorded -- a replay after data.edge_index.copy_(new_batch) then runs them on the new contents
        layers.drop_input_caches()
        try:
            with torch.cuda.graph(self.graph):
                self.loss = step()
            seen, extra = set(id(p) for p in self.params), []
            for t in flags.RAW_WRITTEN:               # BatchNorm running statistics / counters the captured kernels update in place
                if id(t) not in seen:
                    seen.add(id(t))
                    extra.append(t)
            self._bump = self.params + extra
        finally:
            flags.RAW_WRITTEN = None
        # what the capture cached -- derived weights keyed on the parameters' versions, indices of the inputs -- lives in graph-pool memory
        # that nothing has written yet (a capture records): an eager forward before the first replay must not find it
        layers.drop_input_caches()
        layers.drop_capture_caches()
        torch.autograd.graph.increment_version(self._bump)
        self.replays = 0
        self.steps_taken = max(1, warmup)              # (a capture records, it does not execute)

    def __call__(self):
        self.graph.replay()
        torch.autograd.graph.increment_version(self._bump)
        self.replays += 1
        self.steps_taken += 1
        return self.loss

    replay = __call__

"""A whole training step -- forward, loss, zero_grad, backward, Adam: the body of the reference's loop
(pytorch/src/optim_helper.py:156-178) -- captured ONCE into a hipGraph and replayed per batch.

Why: one step is ~250 kernel launches from Python / ctypes / the autograd engine.  At the benchmark grid the GPU
is busy end to end and the launches hide behind it, but on the reference's own training crops (HR 16x64x64 ..
32x64x64, a few ms of GPU work per step) and on the deep U-Net levels the step is launch-bound.  A replay costs one
launch.  This is the MI355X-native stand-in for what a tracing compiler would do, with nothing traced: the captured
kernels are the same hand-written ones, in the same order, on the same stream.

Requirements (checked): ``FlatAdam(capturable=True)`` (step number in device memory), static batch shape, no
gradient all-reduce inside the step (single GPU; the DDP reducer stays on the eager path)."""
from typing import Callable

import torch

from .. import ops
from .optim import FlatAdam


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, loss_fn: Callable, optimizer: FlatAdam, Xs: torch.Tensor,
                 bs: torch.Tensor, ys: torch.Tensor, warmup: int = 2):
        if not isinstance(optimizer, FlatAdam) or not optimizer.capturable:
            raise ValueError("GraphedTrainStep needs FlatAdam(..., capturable=True)")
        if not (Xs.is_cuda and bs.is_cuda and ys.is_cuda):
            raise RuntimeError("GraphedTrainStep: batches must already be on the GPU")
        self.model, self.loss_fn, self.opt = model, loss_fn, optimizer
        self.x, self.b, self.y = Xs.clone(), bs.clone(), ys.clone()      # static input buffers of the graph
        self.loss = None
        opt = optimizer
        # warm-up (lazy kernel attributes, allocator pools) must leave no trace: the steps below are undone
        keep = [t.clone() for t in (opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt._step_dev)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for dst, src in zip((opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt._step_dev), keep):
                dst.copy_(src)
        # the warm-up's blocks go back to the driver: the capture allocates from its own pool, and at BASELINE configs[4]
        # (160 GB of bf16 activations) two cached copies of a step's working set would not fit the 288 GB
        del keep
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: the capture happens on the first batch INSIDE the training loop, while the DataLoader's pin-memory
        # thread may be calling hipHostMalloc / hipEventQuery for the batches it prefetches; in the default "global" mode
        # such a call from another thread invalidates the capture.  Only this thread's calls are part of the graph.
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = self._body()
        self.replays = 0

    def _body(self) -> torch.Tensor:
        pred = self.model(self.x, self.b)
        loss = self.loss_fn(pred, self.y, self.b)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss.detach()

    def __call__(self, Xs: torch.Tensor, bs: torch.Tensor, ys: torch.Tensor) -> torch.Tensor:
        """one training step on this batch; returns the (device) loss of the batch -- overwritten by the next call"""
        if Xs.shape != self.x.shape or bs.shape != self.b.shape or ys.shape != self.y.shape:
            raise ValueError(f"GraphedTrainStep was captured for batches {tuple(self.x.shape)} / {tuple(self.b.shape)} "
                             f"/ {tuple(self.y.shape)}")
        self.x.copy_(Xs, non_blocking=True)
        self.b.copy_(bs, non_blocking=True)
        self.y.copy_(ys, non_blocking=True)
        self.graph.replay()
        ops.invalidate_eval_cache()     # the replay rewrote buffers the metrics cache may have seen
        self.replays += 1
        return self.loss

"""A whole training step -- forward, loss, zero_grad, backward, [gradient all-reduce,] Adam: the body of the reference's
loop (pytorch/src/optim_helper.py:156-178) -- captured ONCE into a hipGraph and replayed per batch.

Why: one step is ~250 kernel launches from Python / ctypes / the autograd engine.  At the benchmark grid the GPU
is busy end to end and the launches hide behind it, but on the reference's own training crops (HR 16x64x64 ..
32x64x64, a few ms of GPU work per step) and on the deep U-Net levels the step is launch-bound.  A replay costs one
launch.  This is the MI355X-native stand-in for what a tracing compiler would do, with nothing traced: the captured
kernels are the same hand-written ones, in the same order, on the same stream.

Requirements (checked): ``FlatAdam(capturable=True)`` (step number in device memory), static batch shape.

Data parallel (``reducer`` = the GradAllReducer that stands in for DistributedDataParallel, reference
pytorch/script/train_model.py:179) -- BASELINE configs[4] is "8 GPUs AND a captured step" -- in one of two ways:

``comm="split"`` (default)
    graph A = forward + loss + zero_grad + backward with the reducer's bucket hooks muted; then the bucket all-reduces,
    EAGER, on the reducer's side stream, in the order backward produced them; then Adam (one kernel, the 1/world factor
    folded in).  RCCL only ever sees ordinary stream launches, exactly as on the eager path, so nothing depends on the
    collective library's capture support; what is lost is the overlap of the all-reduce with the rest of backward:
    262 MB over xGMI = 1-3 ms against the >= 0.9 s of a configs[4] step.
``comm="captured"``
    ONE graph: the hooks fire inside the captured backward, the all-reduces are captured on the reducer's side stream
    (forked from / joined to the capturing stream by events), overlap kept.  Needs a collective backend that can be
    stream-captured (RCCL can; gloo cannot) -- opt-in, ``SR3D_GRAPH_COMM=captured``.

On a CPU device (the gloo rehearsal of this control flow, tests/test_dist_paths_gloo.py) a "graph" is a recorded
callable that replay() calls again: same segments, same order of collectives, no HIP."""
import os
from typing import Callable, Optional

import torch

from .. import ops


class _Recorded:
    """CPU stand-in for a captured graph: replay() re-runs the segment (test rehearsal of the control flow only)"""

    def __init__(self, fn: Callable):
        self.fn, self.out = fn, None

    def replay(self):
        self.out = self.fn()


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, loss_fn: Callable, optimizer, Xs: torch.Tensor,
                 bs: torch.Tensor, ys: torch.Tensor, warmup: int = 2, reducer=None, comm: Optional[str] = None):
        if not getattr(optimizer, "capturable", False) or not hasattr(optimizer, "flat_param"):
            raise ValueError("GraphedTrainStep needs FlatAdam(..., capturable=True)")
        self.on_gpu = Xs.is_cuda
        if not (Xs.device == bs.device == ys.device):
            raise RuntimeError("GraphedTrainStep: the batch tensors must live on one device")
        if not self.on_gpu and optimizer.flat_param.is_cuda:
            raise RuntimeError("GraphedTrainStep: batches must already be on the GPU")
        comm = comm or os.environ.get("SR3D_GRAPH_COMM", "split")
        if comm not in ("split", "captured"):
            raise ValueError(f"GraphedTrainStep: comm must be 'split' or 'captured' (got {comm!r})")
        self.model, self.loss_fn, self.opt, self.reducer = model, loss_fn, optimizer, reducer
        self.comm = comm if reducer is not None else None
        self.x, self.b, self.y = Xs.clone(), bs.clone(), ys.clone()      # static input buffers of the graph
        self.loss = None
        opt = optimizer
        # warm-up (lazy kernel attributes, allocator pools, the communicator's first collective) must leave no trace: the
        # steps below are undone.  Every rank runs the same number of them, so the collectives pair up.
        state = [t for t in (opt.flat_param, opt.exp_avg, opt.exp_avg_sq, getattr(opt, "_step_dev", None)) if t is not None]
        keep = [t.clone() for t in state]
        host_step = getattr(opt, "_host_step", None)
        if self.on_gpu:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    self._eager_body()
            torch.cuda.current_stream().wait_stream(side)
        else:
            for _ in range(max(1, warmup)):
                self._eager_body()
        with torch.no_grad():
            for dst, src in zip(state, keep):
                dst.copy_(src)
        if host_step is not None:
            opt._host_step = host_step
        # the warm-up's blocks go back to the driver: the capture allocates from its own pool, and at BASELINE configs[4]
        # (160 GB of bf16 activations) two cached copies of a step's working set would not fit the 288 GB
        del keep
        if self.on_gpu:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        if self.comm == "split":
            self.graph = self._capture(self._fwd_bwd_muted)
        else:
            self.graph = self._capture(self._eager_body)
        self.replays = 0

    # ---- the pieces of the step
    def _fwd_bwd(self) -> torch.Tensor:
        pred = self.model(self.x, self.b)
        loss = self.loss_fn(pred, self.y, self.b)
        self.opt.zero_grad()
        loss.backward()
        return loss.detach()

    def _fwd_bwd_muted(self) -> torch.Tensor:
        self.reducer.hooks_enabled = False          # (no collective inside graph A)
        try:
            return self._fwd_bwd()
        finally:
            self.reducer.hooks_enabled = True

    def _reduce_and_update(self) -> None:
        if self.reducer is not None:
            self.opt.grad_scale = self.reducer.finish()     # launches whatever the hooks did not, waits for all buckets
        self.opt.step()

    def _eager_body(self) -> torch.Tensor:
        """the step exactly as it will run: `split` = muted backward, then the reduces, then Adam"""
        loss = self._fwd_bwd_muted() if self.comm == "split" else self._fwd_bwd()
        self._reduce_and_update()
        return loss

    def _capture(self, fn: Callable):
        if not self.on_gpu:
            rec = _Recorded(fn)
            return rec
        g = torch.cuda.CUDAGraph()
        # thread_local: the capture happens on the first batch INSIDE the training loop, while the DataLoader's pin-memory
        # thread may be calling hipHostMalloc / hipEventQuery for the batches it prefetches; in the default "global" mode
        # such a call from another thread invalidates the capture.  Only this thread's calls are part of the graph.
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self.loss = fn()
        return g

    def __call__(self, Xs: torch.Tensor, bs: torch.Tensor, ys: torch.Tensor) -> torch.Tensor:
        """one training step on this batch; returns the (device) loss of the batch -- overwritten by the next call"""
        if Xs.shape != self.x.shape or bs.shape != self.b.shape or ys.shape != self.y.shape:
            raise ValueError(f"GraphedTrainStep was captured for batches {tuple(self.x.shape)} / {tuple(self.b.shape)} "
                             f"/ {tuple(self.y.shape)}")
        self.x.copy_(Xs, non_blocking=True)
        self.b.copy_(bs, non_blocking=True)
        self.y.copy_(ys, non_blocking=True)
        self.graph.replay()
        if not self.on_gpu:
            self.loss = self.graph.out
        if self.comm == "split":
            self._reduce_and_update()   # eager: bucket all-reduces on the reducer's stream, then the Adam kernel
        ops.invalidate_eval_cache()     # the replay rewrote buffers the metrics cache may have seen
        self.replays += 1
        return self.loss

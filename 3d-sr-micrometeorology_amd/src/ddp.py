"""Data-parallel gradient averaging: bucketed all-reduce over RCCL/xGMI,
overlapped with the backward pass (stands in for DistributedDataParallel at
reference pytorch/script/train_model.py:179).

One process per GPU.  Gradients live in one flat buffer (src/optim.py); it is
cut into buckets in REVERSE parameter order (the order backward produces them:
``last``, ``up1`` ... first).  A post-accumulate hook on every parameter counts
its bucket down; the last arrival launches an asynchronous SUM all-reduce of the
bucket's slice.  ``finish()`` waits for all of them; the 1/world factor is folded into
the fused Adam kernel (``FlatAdam.grad_scale``) so no extra pass touches the
gradients.  With xGMI being point-to-point, few large buckets are better than
DDP's 25 MB default: the default here is 64 MB (5 buckets for the 262 MB model).

The same class runs on the ``gloo`` backend with CPU tensors, which is how the
multi-process tests exercise it without GPUs."""
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, params: Sequence[torch.nn.Parameter], flat_grad: torch.Tensor, offsets: Sequence[int],
                 bucket_bytes: int = 64 << 20, process_group=None):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("GradAllReducer needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.params, self.flat_grad = list(params), flat_grad
        self.offsets = list(offsets)
        # buckets over the flat buffer, walking parameters from last to first
        self.buckets: List[dict] = []
        cur: Optional[dict] = None
        for idx in reversed(range(len(self.params))):
            p, o = self.params[idx], self.offsets[idx]
            end = o + (p.numel() + 3) // 4 * 4
            if cur is None or (cur["end"] - o) * 4 > bucket_bytes:
                cur = {"begin": o, "end": end, "members": 0, "pending": 0, "work": None}
                self.buckets.append(cur)
            cur["begin"] = o
            cur["members"] += 1
            p._sr3d_bucket = len(self.buckets) - 1
        self._handles = []
        for p in self.params:
            self._handles.append(p.register_post_accumulate_grad_hook(self._hook))
        self.comm_stream = torch.cuda.Stream() if flat_grad.is_cuda else None
        # False: the hooks launch nothing and finish() reduces every bucket, in the order backward produced them -- a
        # backward that is being captured into a hipGraph WITHOUT its collectives (src/graph.py, comm="split")
        self.hooks_enabled = True
        self.reset()

    def reset(self) -> None:
        for b in self.buckets:
            b["pending"], b["work"] = b["members"], None

    def broadcast_parameters(self, flat_param: torch.Tensor, src: int = 0) -> None:
        """initial weight sync (DDP constructor broadcast, train_model.py:179)"""
        dist.broadcast(flat_param, src=src, group=self.group)

    def _launch(self, b: dict) -> None:
        chunk = self.flat_grad[b["begin"]:b["end"]]
        if self.comm_stream is not None:
            # RCCL runs on its own stream: order it after the kernels that produced this bucket
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                b["work"] = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b["work"] = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _hook(self, p: torch.nn.Parameter) -> None:
        if not self.hooks_enabled:
            return
        b = self.buckets[p._sr3d_bucket]
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def finish(self) -> float:
        """wait for every bucket; returns the factor that turns the SUM into the mean"""
        for b in self.buckets:
            if b["work"] is None:  # parameter without gradient this step: reduce what is there (zeros)
                self._launch(b)
            b["work"].wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.reset()
        return 1.0 / self.world

    def remove_hooks(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []

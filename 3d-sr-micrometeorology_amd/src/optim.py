"""Fused Adam over ONE flat fp32 buffer (replaces torch.optim.Adam of reference
pytorch/script/train_model.py:183 with the same defaults).

All parameters are re-pointed to views of a single contiguous buffer, their
``.grad`` to views of a second one, so the optimizer step is a single streaming
HIP kernel (28 B/param) and the data-parallel all-reduce works on contiguous
buckets of the same buffer (src/ddp.py)."""
from typing import Iterable, List

import torch

from .. import ops


def flatten_parameters(params: Iterable[torch.nn.Parameter]):
    """Re-point every trainable parameter to a view of ONE contiguous fp32 buffer and its ``.grad`` to a view of a
    second one.  Offsets are multiples of 4 elements, so every view can be processed with 128-bit accesses.
    Device-agnostic host logic (the gloo tests lay out the real 48-tensor model on the CPU with it).
    Returns (params, offsets, flat_param, flat_grad)."""
    plist: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
    if not plist:
        raise ValueError("flatten_parameters: no trainable parameters")
    dev = plist[0].device
    offsets, off = [], 0
    for p in plist:
        if p.device != dev or p.dtype != torch.float32:
            raise ValueError("flatten_parameters: parameters must be fp32 tensors on one device")
        offsets.append(off)
        off += (p.numel() + 3) // 4 * 4
    flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
    flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
    with torch.no_grad():
        for p, o in zip(plist, offsets):
            view = flat_param[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = flat_grad[o:o + p.numel()].view_as(p)
    return plist, offsets, flat_param, flat_grad


class FlatAdam:
    """``capturable=True`` keeps the step number in device memory (incremented by the optimizer kernel itself), which
    is what lets a whole training step be captured into a hipGraph and replayed (src/graph.py)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 capturable: bool = False):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FlatAdam: no trainable parameters")
        if params[0].device.type != "cuda":
            raise RuntimeError("FlatAdam runs on the GPU only (move the model with .to('cuda') first)")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self._host_step = 0
        self.capturable = bool(capturable)
        self.grad_scale = 1.0  # e.g. 1/world_size when gradients were SUM-all-reduced
        self.params, self.offsets, self.flat_param, self.flat_grad = flatten_parameters(params)
        self.numel = self.flat_param.numel()
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self._step_dev = torch.zeros(1, dtype=torch.int32, device=self.flat_param.device) if self.capturable else None
        self._scalars = torch.zeros(2, dtype=torch.float32, device=self.flat_param.device) if self.capturable else None

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):  # keep the views alive if someone dropped them
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    @property
    def step_count(self) -> int:
        return int(self._step_dev.item()) if self.capturable else self._host_step

    @torch.no_grad()
    def step(self) -> None:
        if self.capturable:
            ops.adam_step_device_counter_(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.lr,
                                          self.betas[0], self.betas[1], self.eps, self._step_dev, self._scalars,
                                          self.grad_scale)
            return
        self._host_step += 1
        ops.adam_step_(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                       self.betas[1], self.eps, self._host_step, self.grad_scale)

    def state_dict(self) -> dict:
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr,
                "betas": self.betas, "eps": self.eps}

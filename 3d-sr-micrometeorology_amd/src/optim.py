"""Fused Adam over ONE flat fp32 buffer (replaces torch.optim.Adam of reference
pytorch/script/train_model.py:183 with the same defaults).

All parameters are re-pointed to views of a single contiguous buffer, their
``.grad`` to views of a second one, so the optimizer step is a single streaming
HIP kernel (28 B/param) and the data-parallel all-reduce works on contiguous
buckets of the same buffer (src/ddp.py)."""
from typing import Iterable, List

import torch

from .. import ops


class FlatAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam: no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam runs on the GPU only (move the model with .to('cuda') first)")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.step_count = 0
        self.grad_scale = 1.0  # e.g. 1/world_size when gradients were SUM-all-reduced
        # 16-byte aligned offsets so every view can be processed with 128-bit accesses
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_param[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):  # keep the views alive if someone dropped them
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    @torch.no_grad()
    def step(self) -> None:
        self.step_count += 1
        ops.adam_step_(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                       self.betas[1], self.eps, self.step_count, self.grad_scale)

    def state_dict(self) -> dict:
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr,
                "betas": self.betas, "eps": self.eps}

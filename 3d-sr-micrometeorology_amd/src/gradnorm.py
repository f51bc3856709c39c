"""GradNorm (Chen et al., ICML 2018) loss balancing behind the interface the reference's trainers call
(pytorch/src/gradnorm.py:13-115 -- constructor keywords, ``weights``, ``init_losses``, ``backward``,
``renormalize_weights``, ``calc_total_weighted_loss_for_test``, ``record_and_write_out_weights_and_losses``;
call sites pytorch/src/optim_helper.py:44-60,96-99,167-178,216-219 and script/train_model.py:185-199,228-236).

Written from the algorithm, not from the reference's text:

  task i has loss L_i(t) and a positive weight w_i;  the model minimises  sum_i w_i L_i.
  G_i   = || d(w_i L_i) / dW ||_2 = |w_i| * n_i,    n_i = || dL_i/dW ||_2,   W = first tensor of the last layer
  r_i   = (L_i(t) / L_i(0)) / mean_j (L_j(t) / L_j(0))                      relative inverse training rate
  c_i   = mean_j(G_j) * r_i ** alpha                                         target norm, treated as a constant
  L_gn  = sum_i | G_i - c_i |                                                minimised w.r.t. w only
  dL_gn/dw_i = sign(G_i - c_i) * sign(w_i) * n_i                             (closed form; no second autograd graph)

The per-task norms n_i are the only expensive part: one backward per task from its loss term to W.  On this engine that
is ``ops.MixedLossFn``'s adjoint-stencil kernel plus the ``last`` convolution's weight-gradient kernel, launched by
``torch.autograd.grad``; everything else here is arithmetic on n_tasks-vectors that stay on the device.
"""
from __future__ import annotations

import os
from logging import getLogger
from typing import List, Optional, Sequence

import torch

logger = getLogger()


class GradNorm:
    def __init__(self, n_tasks: int, alpha: float = 1.5, device: Optional[str] = None, output_dir_path: str = ".",
                 clipping_weight_min: Optional[float] = None, **kwargs):
        self.n_tasks = int(n_tasks)
        self.alpha = float(alpha)
        self.device = device
        self.dir_path = output_dir_path
        self.clipping_min = clipping_weight_min
        # leaf tensor: the caller puts it into its own optimizer (train_model.py:191-199) and steps it after backward()
        self.weights = torch.ones(self.n_tasks, device=device, requires_grad=True)
        self.init_losses: Optional[torch.Tensor] = None      # L_i(0), taken from the first training batch
        self._eval_terms: List[torch.Tensor] = []            # loss terms seen by the validation loop this epoch
        self._weight_history: List[List[float]] = []
        self._loss_history: List[List[float]] = []
        logger.info(f"GradNorm: {self.n_tasks} tasks, alpha = {self.alpha}, weight floor = {self.clipping_min}")

    # ------------------------------------------------------------------ training step
    def backward(self, loss_list: Sequence[torch.Tensor], last_shared_params: Sequence[torch.nn.Parameter],
                 return_total_weighted_loss: bool = True, **kwargs):
        """Accumulates d(sum_i w_i L_i)/d(theta) into the model parameters' ``.grad`` and puts the GradNorm gradient
        into ``self.weights.grad``.  Returns the weighted total (a tensor attached to the graph, like the reference)."""
        terms = torch.stack(list(loss_list))
        if terms.numel() != self.n_tasks:
            raise ValueError(f"GradNorm was built for {self.n_tasks} tasks, got {terms.numel()} loss terms")
        if self.init_losses is None:
            self.init_losses = terms.detach().clone()
        w = self.weights
        W = last_shared_params[0]

        # n_i: one backward per task down to the shared last layer (graph kept: the model backward comes after)
        n = torch.stack([torch.autograd.grad(terms[i], W, retain_graph=True)[0].norm() for i in range(self.n_tasks)])

        total = (w.detach() * terms).sum()
        total.backward()                    # model gradients only: the weights enter as constants

        with torch.no_grad():
            G = w.abs() * n
            rel = terms.detach() / self.init_losses
            r = rel / rel.mean()
            target = G.mean() * r.pow(self.alpha)
            self.weights.grad = torch.sign(G - target) * torch.sign(w) * n
        if return_total_weighted_loss:
            return total

    def renormalize_weights(self) -> None:
        """after the weight optimizer's step: optional floor, then rescale so that sum_i w_i = n_tasks"""
        with torch.no_grad():
            if self.clipping_min is not None:
                self.weights.clamp_(min=self.clipping_min)
            self.weights.mul_(self.n_tasks / self.weights.sum())
        self.weights.requires_grad_(True)

    # ------------------------------------------------------------------ validation / bookkeeping
    def calc_total_weighted_loss_for_test(self, loss_list: Sequence[torch.Tensor]) -> torch.Tensor:
        with torch.no_grad():
            terms = torch.stack(list(loss_list)).detach()
            self._eval_terms.append(terms)
            return (self.weights.detach() * terms).sum()

    def record_and_write_out_weights_and_losses(self) -> None:
        """once per epoch: append the current weights and the epoch-mean validation terms to two CSV files"""
        import pandas as pd
        self._weight_history.append(self.weights.detach().cpu().tolist())
        if self._eval_terms:
            self._loss_history.append(torch.stack(self._eval_terms).mean(dim=0).cpu().tolist())
            self._eval_terms = []
        pd.DataFrame(self._weight_history).to_csv(os.path.join(self.dir_path, f"grad_norm_weights_{self.device}.csv"))
        pd.DataFrame(self._loss_history).to_csv(os.path.join(self.dir_path, f"grad_norm_losses_{self.device}.csv"))

    # names the reference exposes for its recorded histories
    @property
    def recorded_weights(self):
        return self._weight_history

    @property
    def recorded_losses(self):
        return self._loss_history

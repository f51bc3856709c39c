"""GradNorm adaptive loss weighting with the reference's interface
(pytorch/src/gradnorm.py:13-115).  The per-term gradients it takes w.r.t. the
last layer (``torch.autograd.grad(L_i, last_shared_params)``, :95-100) run
through the engine's differentiable loss terms (``ops.MixedLossFn``) and the
``last`` conv's hand-written weight-gradient kernel."""
from copy import deepcopy
from logging import getLogger
from typing import List

import numpy as np
import torch

logger = getLogger()


class GradNorm:
    def __init__(self, n_tasks: int, alpha: float = 1.5, device: str = None, output_dir_path: str = ".",
                 clipping_weight_min: float = None, **kwargs):
        self.n_tasks = n_tasks
        self.alpha = alpha
        self.weights = torch.ones((n_tasks,), requires_grad=True, device=device)
        self.dir_path = output_dir_path
        self.device = device
        self.clipping_min = clipping_weight_min
        self.init_losses = None
        self._losses = []
        self.recorded_weights = []
        self.recorded_losses = []
        logger.info(f"GradNorm params: n_tasks = {self.n_tasks}, alpha = {self.alpha}, "
                    f"clipping_weight_min = {self.clipping_min}")

    def renormalize_weights(self):
        with torch.no_grad():
            if self.clipping_min is not None:
                self.weights = self.weights.clamp_(min=self.clipping_min)
            self.weights *= self.n_tasks / self.weights.sum()
        self.weights.requires_grad = True

    def calc_total_weighted_loss_for_test(self, loss_list: List[torch.Tensor]):
        with torch.no_grad():
            losses = torch.stack(loss_list)
            self._losses.append(losses.detach().cpu().numpy())
            return (self.weights * losses).sum()

    def record_and_write_out_weights_and_losses(self):
        import pandas as pd
        self.recorded_weights.append(deepcopy(self.weights.detach().cpu().numpy()))
        self.recorded_losses.append(np.mean(np.stack(self._losses, axis=0), axis=0))
        self._losses = []
        pd.DataFrame(self.recorded_weights).to_csv(f"{self.dir_path}/grad_norm_weights_{self.device}.csv")
        pd.DataFrame(self.recorded_losses).to_csv(f"{self.dir_path}/grad_norm_losses_{self.device}.csv")

    def backward(self, loss_list: List[torch.Tensor], last_shared_params: List[torch.nn.Parameter],
                 return_total_weighted_loss: bool = True, **kwargs):
        losses = torch.stack(loss_list)
        if self.init_losses is None:
            self.init_losses = losses.detach().clone()
        total_weighted_loss = (self.weights * losses).sum()
        total_weighted_loss.backward(retain_graph=True)
        # the backward above also produced d(total)/d(weights); GradNorm sets that gradient itself below
        self.weights.grad = torch.zeros_like(self.weights.grad)

        norms = []
        for w_i, L_i in zip(self.weights, losses):
            grd_L_i = torch.autograd.grad(L_i, last_shared_params, retain_graph=True)[0]
            norms.append(torch.norm(w_i * grd_L_i))
        norms = torch.stack(norms)

        with torch.no_grad():
            loss_ratios = losses / self.init_losses
            inverse_train_rates = loss_ratios / loss_ratios.mean()
            constant_term = (norms.mean() * (inverse_train_rates ** self.alpha)).detach().clone()

        grad_norm_loss = (norms - constant_term).abs().sum()
        self.weights.grad = torch.autograd.grad(grad_norm_loss, self.weights)[0]
        if return_total_weighted_loss:
            return total_weighted_loss

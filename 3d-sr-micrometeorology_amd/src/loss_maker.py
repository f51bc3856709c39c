"""Training losses with the reference's factory and module interface
(pytorch/src/loss_maker.py:19-54, 194-213, 358-450), evaluated by the fused HIP
loss kernels (value and dL/dprediction in one launch sequence)."""
from copy import deepcopy
from logging import getLogger
from typing import List

import numpy as np
import torch
from torch import nn

from .. import ops

logger = getLogger()


def make_loss(config: dict) -> nn.Module:
    name = config["train"]["loss"]["name"]
    if name == "L1":
        logger.info("L1 loss is created.")
        return MyL1Loss()
    if name == "MixedDivergenceGradientL2Loss":
        logger.info("MixedDivergenceGradientL2Loss is created")
        return MixedDivergenceGradientL2Loss(
            weight_gradient_loss=config["train"]["loss"].get("weight_gradient_loss", 0.0),
            weight_divergence_loss=config["train"]["loss"].get("weight_divergence_loss", 0.0),
            scales=config["data"]["stds"][1:],
        )
    if name == "L2":
        logger.info("L2 loss is created.")
        return MyL2Loss()
    # WeightedL1 / WeightedL2 / MixedGradientL2Loss (loss_maker.py:27-38) are not on the hot path
    # named by BASELINE.json and have no fused kernel yet.
    raise NotImplementedError(f"{name} is not supported.")


def calc_mask_near_build_wall(building: torch.Tensor, num_filter_applications: int = 1) -> torch.Tensor:
    """loss_maker.py:57-83 (one filter application, the only value the reference uses)."""
    assert len(building.shape) == 5
    if num_filter_applications != 1:
        raise NotImplementedError("num_filter_applications != 1")
    return ops.near_wall_mask(building)


class MyL1Loss(nn.Module):
    def forward(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor = None):
        return ops.L1LossFn.apply(predicts, targets)


class MyL2Loss(nn.Module):
    """mean squared error = the first term of the mixed loss with both weights zero."""

    def forward(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor):
        terms = ops.MixedLossFn.apply(predicts, targets, masks, [1.0, 1.0, 1.0], 5.0, 0.0, 0.0)
        return terms[3]


class MixedDivergenceGradientL2Loss(nn.Module):
    def __init__(self, weight_gradient_loss: float, weight_divergence_loss: float, scales: List[float],
                 delta_meter: float = 5.0):
        super().__init__()
        assert len(scales) == 3, "velocity components have 3. So scales length must be 3."
        self.weight_gradient_loss = weight_gradient_loss
        self.weight_divergence_loss = weight_divergence_loss
        self.scales = deepcopy(scales)
        self.mean_scale = np.mean(scales)
        self.delta_meter = delta_meter
        logger.info(f"weight grad loss = {self.weight_gradient_loss}")
        logger.info(f"weight divergence loss = {self.weight_divergence_loss}")
        logger.info(f"velocity scales = {self.scales}, its mean = {self.mean_scale}")
        logger.info(f"delta meter = {self.delta_meter}")

    def _terms(self, predicts, targets, masks) -> torch.Tensor:
        return ops.MixedLossFn.apply(predicts, targets, masks, self.scales, self.delta_meter,
                                     self.weight_gradient_loss, self.weight_divergence_loss)

    def calc_loss_terms(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor):
        """(mse, grd_mse, div_mse); skipped terms are the float 0.0 as in loss_maker.py:399-413.
        Each returned tensor is differentiable on its own (GradNorm: optim_helper.py:167-175)."""
        t = self._terms(predicts, targets, masks)
        grd = t[1] if self.weight_gradient_loss != 0.0 else 0.0
        div = t[2] if self.weight_divergence_loss != 0.0 else 0.0
        return t[0], grd, div

    def forward(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor):
        return self._terms(predicts, targets, masks)[3]

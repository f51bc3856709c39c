"""Training losses and evaluation metrics with the reference's factory and module interface
(pytorch/src/loss_maker.py:19-54, 194-213, 358-450 and 453-745).  Training losses run the fused HIP loss kernels
(value and dL/dprediction in one launch sequence); the evaluation metrics all read their value from ONE fused
pass over (prediction, target, mask) (``ops.eval_metrics``), however many of them are evaluated on a batch."""
from copy import deepcopy
from logging import getLogger
import typing
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
    if name == "WeightedL1":
        logger.info("Weighted L1 loss is created.")
        return WeightedL1Loss(config["train"]["loss"]["weight_outside_building"])
    if name == "WeightedL2":
        logger.info("Weighted L2 loss is created.")
        return WeightedL2Loss(config["train"]["loss"]["weight_outside_building"])
    if name == "MixedGradientL2Loss":
        logger.info("Mixed gradient L2 loss is created.")
        return MixedGradientL2Loss(weight_gradient_loss=config["train"]["loss"].get("weight_gradient_loss", None))
    raise NotImplementedError(f"{name} is not supported.")


def calc_mask_near_build_wall(building: torch.Tensor, num_filter_applications: int = 1) -> torch.Tensor:
    """loss_maker.py:57-83 (one filter application, the only value the reference uses)."""
    assert len(building.shape) == 5
    if num_filter_applications != 1:
        raise NotImplementedError("num_filter_applications != 1")
    return ops.near_wall_mask(building)


class MyL1Loss(nn.Module):
    def forward(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor = None):
        if not (torch.is_grad_enabled() and predicts.requires_grad) and ops.eval_shapes_ok(predicts, targets, masks):
            # evaluation: the value comes out of the fused metrics pass (shared with the other metrics of the batch)
            return ops.eval_metrics(predicts, targets, masks, (None,) * 4)[0]
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


class WeightedL1Loss(nn.Module):
    """loss_maker.py:216-232"""
    power = 1

    def __init__(self, weight_outside_building: float):
        super().__init__()
        self.weight = weight_outside_building

    def forward(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor):
        if ops.eval_shapes_ok(predicts, targets, masks):
            return ops.WeightedLpFn.apply(predicts, targets, masks, float(self.weight), self.power)
        # any other broadcastable shape (the reference accepts them, loss_maker.py:222-232): the plain expressions
        e = (predicts - targets).abs() if self.power == 1 else (predicts - targets) ** 2
        inside = torch.broadcast_to(masks, e.shape)
        n_in = inside.sum()
        n_out = e.numel() - n_in
        s_in = (e * inside).sum()
        return (self.weight * s_in / (n_in + 1.0) + (e.sum() - s_in) / (n_out + 1.0)) / (self.weight + 1.0)


class WeightedL2Loss(WeightedL1Loss):
    """loss_maker.py:235-255"""
    power = 2


class MixedGradientL2Loss(nn.Module):
    """loss_maker.py:258-301: mse + w_g * grd_mse -- the mixed divergence-gradient loss without its divergence term
    (same masks, same 4 * sum(M) + 1 denominator)"""

    def __init__(self, weight_gradient_loss: float):
        super().__init__()
        self.weight_gradient_loss = weight_gradient_loss
        logger.info(f"weight grad loss = {self.weight_gradient_loss}")

    def _off(self) -> bool:
        return self.weight_gradient_loss is None or self.weight_gradient_loss == 0

    def calc_loss_terms(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor):
        t = ops.MixedLossFn.apply(predicts, targets, masks, [1.0, 1.0, 1.0], 5.0,
                                  0.0 if self._off() else float(self.weight_gradient_loss), 0.0)
        return (t[0], None) if self._off() else (t[0], t[1])

    def forward(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor):
        mse, grd = self.calc_loss_terms(predicts=predicts, targets=targets, masks=masks)
        return mse if self._off() else mse + self.weight_gradient_loss * grd


class MixedGradientWeightedL2Loss(nn.Module):
    """loss_maker.py:304-355 (no ``make_loss`` branch selects it; constructor and methods as in the reference):
    (w * mse_fluid + mse_buildings) / (w + 1) + w_g * grd_mse.  The first part is ``WeightedL2Loss`` (five sums of the fused
    evaluation pass + one streaming gradient kernel), the gradient term the second entry of the fused mixed-loss kernel --
    the reference's 4-channel mask sums to the same ``4 * sum(M) + 1`` denominator."""

    def __init__(self, weight_outside_building: float, weight_gradient_loss: float):
        super().__init__()
        self.weight_outside_building = weight_outside_building
        self.weight_gradient_loss = weight_gradient_loss

    def _grd(self, predicts, targets, masks):
        return ops.MixedLossFn.apply(predicts, targets, masks, [1.0, 1.0, 1.0], 5.0, 1.0, 0.0)[1]

    def calc_loss_terms(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor):
        """(one_region_diff, zero_region_diff, grd_mse), each differentiable"""
        one = ops.WeightedLpFn.apply(predicts, targets, masks, 1.0e18, 2)       # w -> inf: the fluid voxels' average alone
        zero = ops.WeightedLpFn.apply(predicts, targets, masks, 0.0, 2)          # w = 0: the buildings' average alone
        return one, zero, self._grd(predicts, targets, masks)

    def forward(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor):
        mse = ops.WeightedLpFn.apply(predicts, targets, masks, float(self.weight_outside_building), 2)
        return mse + self.weight_gradient_loss * self._grd(predicts, targets, masks)


class ChannelwiseMse(nn.Module):
    """loss_maker.py:753-764: mean squared error of ONE of the four channels (the mixed kernel's mse term on the slice)"""

    def __init__(self, i_channel):
        super().__init__()
        self.i_channel = i_channel

    def forward(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor):
        assert predicts.shape[1] == targets.shape[1] == 4
        i = self.i_channel
        d = predicts[:, i] - targets[:, i]      # (one channel of four: a strided slice; plain ATen on the device)
        return (d * d).mean()


# ---------------------------------------------------------------------------------------------------------------
# evaluation metrics (no gradient): pytorch/src/loss_maker.py:453-745, used by script/train_model.py:366-390 and
# the evaluation notebooks.  Constructor arguments and forward(predicts, targets, masks) as in the reference.
from .._lib import EVAL_INDEX  # noqa: E402


class _FusedMetric(nn.Module):
    """one entry of the fused evaluation pass"""
    entry: str = ""

    def _stds(self):           # (std_T, std_u, std_v, std_w) as far as this metric depends on them
        return (None, None, None, None)

    def _lev(self) -> int:
        return 0

    delta_meter = 5.0

    def forward(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor):
        out = ops.eval_metrics(predicts, targets, masks, self._stds(), self.delta_meter, self._lev())
        return out[EVAL_INDEX[self.entry]]


def _check_eps(eps: float):
    if eps != 1e-30:
        raise NotImplementedError("the fused evaluation kernel implements the reference's default eps = 1e-30")


class MaskedL1Loss(_FusedMetric):
    entry = "MaskedL1"

    def __init__(self, eps: float = 1e-30):
        super().__init__()
        _check_eps(eps)
        self.eps = eps


class MaskedL2Loss(MaskedL1Loss):
    entry = "MaskedL2"


class MaskedL1LossNearWall(_FusedMetric):
    entry = "MaskedL1NearWall"

    def __init__(self, eps: float = 1e-30, num_filter_applications: int = 1):
        super().__init__()
        _check_eps(eps)
        if num_filter_applications != 1:
            raise NotImplementedError("num_filter_applications != 1")
        self.eps, self.num_filter_applications = eps, num_filter_applications


class MaskedL2LossNearWall(MaskedL1LossNearWall):
    entry = "MaskedL2NearWall"


class _VelocityMetric(_FusedMetric):
    def __init__(self, scales: List[float], delta_meter: float = 5.0):
        super().__init__()
        assert len(scales) == 3
        self.scales = deepcopy(scales)
        self.delta_meter = delta_meter

    def _stds(self):
        return (None, *[float(v) for v in self.scales])


class ResidualContinuity(_VelocityMetric):
    entry = "ResidualContinuity"

    def calc_both_pred_and_target(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor):
        out = ops.eval_metrics(predicts, targets, masks, self._stds(), self.delta_meter, 0)
        return out[EVAL_INDEX["ResidualContinuity"]], out[EVAL_INDEX["ResidualContinuityTarget"]]


class AbsDiffDivergence(_VelocityMetric):
    entry = "AbsDiffDivergence"


class DiffOmegaVectorNorm(_VelocityMetric):
    entry = "DiffOmegaNorm"

    def __init__(self, scales: List[float], delta_meter: float = 5.0):
        super().__init__(scales, delta_meter)
        self.delta = delta_meter


class DiffVelocityVectorNorm(_FusedMetric):
    def __init__(self, scales: List[float], eps: float = 1e-30, lev: int = None):
        super().__init__()
        assert len(scales) == 3
        _check_eps(eps)
        self.scales, self.eps, self.lev = deepcopy(scales), eps, lev
        self.entry = "DiffVelocityNorm" if lev is None else "DiffVelocityNormLev"

    def _stds(self):
        return (None, *[float(v) for v in self.scales])

    def _lev(self):
        return 0 if self.lev is None else int(self.lev)

    def forward(self, predicts, targets, masks):
        assert predicts.shape[1] == targets.shape[1] == 4  # channels == T, u, v, w
        return super().forward(predicts, targets, masks)


class AbsDiffTemperature(_FusedMetric):
    def __init__(self, scale: float, eps: float = 1e-30, lev: int = None):
        super().__init__()
        _check_eps(eps)
        self.scale, self.eps, self.lev = scale, eps, lev
        self.entry = "AbsDiffTemperature" if lev is None else "AbsDiffTemperatureLev"

    def _stds(self):
        return (float(self.scale), None, None, None)

    def _lev(self):
        return 0 if self.lev is None else int(self.lev)

    def forward(self, predicts, targets, masks):
        assert predicts.shape[1] == targets.shape[1] == 4  # channels == T, u, v, w
        return super().forward(predicts, targets, masks)


class _MixedTerm(MixedDivergenceGradientL2Loss):
    """loss_maker.py:453-519: one term of the mixed loss as a metric"""
    term, wg, wd = 0, 0.0, 0.0

    def __init__(self, scales: List[float], delta_meter: float = 5.0):
        super().__init__(weight_gradient_loss=self.wg, weight_divergence_loss=self.wd, scales=scales,
                         delta_meter=delta_meter)

    def forward(self, predicts, targets, masks):
        return self._terms(predicts, targets, masks)[self.term]


class MixedDivergenceGradientL2LossMse(_MixedTerm):
    term, wg, wd = 0, 0.0, 0.0


class MixedDivergenceGradientL2LossGrdMse(_MixedTerm):
    term, wg, wd = 1, 1.0, 0.0


class MixedDivergenceGradientL2LossDivMse(_MixedTerm):
    term, wg, wd = 2, 0.0, 1.0


class Ssim3dLoss(nn.Module):
    """loss_maker.py:748-777 (note its default eps = 1e-3, unlike SSIM3D's 1e-7)"""

    def __init__(self, window_size: int = 11, sigma: float = 1.5, size_average: bool = True, max_val: float = 1.0,
                 eps: float = 1e-3, use_gaussian=True):
        super().__init__()
        from .ssim import SSIM3D
        self.ssim = SSIM3D(window_size=window_size, sigma=sigma, size_average=size_average, max_val=max_val, eps=eps,
                           use_gaussian=use_gaussian)

    def forward(self, predicts: torch.Tensor, targets: torch.Tensor, masks: torch.Tensor):
        assert predicts.shape == targets.shape
        # the reference broadcasts the mask to the prediction's shape; the kernel reads the 1-channel mask directly
        from .ssim import _taps
        m = self.ssim
        return ops.ssim3d(predicts, targets, masks, _taps(m.window_size, m.sigma, m.use_gaussian), m.max_val, m.eps,
                          m.size_average)


def merged_metric_scales(loss_fns) -> typing.Tuple[typing.Optional[float], ...]:
    """union of the scales the fused metrics in ``loss_fns`` depend on (None where nobody cares or they disagree)"""
    merged: typing.List[typing.Optional[float]] = [None] * 4
    clash = [False] * 4
    for fn in loss_fns:
        if isinstance(fn, _FusedMetric):
            for i, v in enumerate(fn._stds()):
                if v is None:
                    continue
                if merged[i] is not None and merged[i] != v:
                    clash[i] = True
                merged[i] = v
    return tuple(None if c else m for m, c in zip(merged, clash))

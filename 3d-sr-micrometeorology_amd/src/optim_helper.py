"""Train / validation loops with the reference's signatures
(pytorch/src/optim_helper.py:22-225).  The step body is unchanged; ``optimizer``
is any object with ``zero_grad()/step()`` (``FlatAdam`` or a torch optimizer) and,
for the DDP variants, gradient averaging is the engine's ``GradAllReducer``
(``reducer`` argument) instead of a ``DistributedDataParallel`` wrapper."""
import typing
from logging import getLogger

import torch
import torch.distributed as dist
from torch import nn

logger = getLogger()


class AverageMeter:
    """running average (pytorch/src/utils.py:52-67)"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def _last_params(model):
    return (model.module if hasattr(model, "module") else model).get_last_params()


def _step(model, loss_fn, optimizer, Xs, bs, ys, grad_norm, reducer=None):
    preds = model(Xs, bs)
    if grad_norm is None:
        loss = loss_fn(preds, ys, bs)
        optimizer.zero_grad()
        loss.backward()
    else:
        losses = loss_fn.calc_loss_terms(predicts=preds, targets=ys, masks=bs)
        optimizer.zero_grad()
        loss = grad_norm.backward(loss_list=list(losses), last_shared_params=_last_params(model))
    if reducer is not None:
        optimizer.grad_scale = reducer.finish()
    optimizer.step()
    if grad_norm is not None:
        grad_norm.renormalize_weights()
    return loss


def train(dataloader, model: nn.Module, loss_fn, optimizer, device: str, num_loops: int = 1,
          hide_progress_bar: bool = True, grad_norm=None) -> float:
    """optim_helper.py:22-66"""
    train_loss = AverageMeter()
    model.train()
    for _ in range(num_loops):
        for Xs, bs, ys in dataloader:
            bs = bs.unsqueeze(1)  # add channel dim
            Xs, bs, ys = Xs.to(device), bs.to(device), ys.to(device)
            loss = _step(model, loss_fn, optimizer, Xs, bs, ys, grad_norm)
            train_loss.update(loss.item(), n=len(Xs))
    logger.info(f"Train error: avg loss = {train_loss.avg:.8f}")
    return train_loss.avg


def test(dataloader, model: nn.Module, loss_fn, device: str, num_loops: int = 1, hide_progress_bar: bool = True,
         grad_norm=None) -> float:
    """optim_helper.py:69-108"""
    val_loss = AverageMeter()
    model.eval()
    with torch.no_grad():
        for _ in range(num_loops):
            for Xs, bs, ys in dataloader:
                bs = bs.unsqueeze(1)
                Xs, bs, ys = Xs.to(device), bs.to(device), ys.to(device)
                preds = model(Xs, bs)
                if grad_norm is None:
                    loss = loss_fn(preds, ys, bs)
                else:
                    losses = loss_fn.calc_loss_terms(predicts=preds, targets=ys, masks=bs)
                    loss = grad_norm.calc_total_weighted_loss_for_test(list(losses))
                val_loss.update(loss.item(), n=len(Xs))
    logger.info(f"Valid error: avg loss = {val_loss.avg:.8f}")
    return val_loss.avg


def evaluate(*, dataloader, model: nn.Module, loss_fns: typing.Dict[str, typing.Callable], device: str,
             hide_progress_bar: bool = True) -> dict:
    """optim_helper.py:111-134"""
    from .. import ops
    from .loss_maker import _FusedMetric, merged_metric_scales
    dict_loss = {k: AverageMeter() for k in loss_fns.keys()}
    fused = [fn for fn in loss_fns.values() if isinstance(fn, _FusedMetric)]
    scales = merged_metric_scales(fused)
    with torch.no_grad():
        for Xs, bs, ys in dataloader:
            bs = bs.unsqueeze(1)
            Xs, bs, ys = Xs.to(device), bs.to(device), ys.to(device)
            preds = model(Xs, bs)
            if fused:   # ONE pass over (preds, ys, bs) with the union of the scales: every fused metric below hits it
                ops.eval_metrics(preds, ys, bs, scales, fused[0].delta_meter, 0)
            vals = {name: fn(preds, ys, bs) for name, fn in loss_fns.items()}
            for name, v in vals.items():     # one host sync per metric only after everything is enqueued
                dict_loss[name].update(v.item(), n=len(Xs))
    return dict_loss


def train_ddp(dataloader, sampler, model: nn.Module, loss_fn, optimizer, epoch: int, rank: int, world_size: int,
              num_loops: int, grad_norm=None, reducer=None) -> float:
    """optim_helper.py:137-183; the loss stays on the device until the epoch-end all-reduce"""
    mean_loss, cnt = 0.0, 0
    sampler.set_epoch(epoch)
    model.train()
    for _ in range(num_loops):
        for Xs, bs, ys in dataloader:
            bs = bs.unsqueeze(1)
            Xs, bs, ys = Xs.to(rank), bs.to(rank), ys.to(rank)
            loss = _step(model, loss_fn, optimizer, Xs, bs, ys, grad_norm, reducer)
            mean_loss += loss.detach() * Xs.shape[0]
            cnt += Xs.shape[0]
    mean_loss /= cnt
    dist.all_reduce(mean_loss, op=dist.ReduceOp.SUM)
    return mean_loss.item() / world_size


def test_ddp(dataloader, sampler, model: nn.Module, loss_fn, epoch: int, rank: int, world_size: int, num_loops: int,
             grad_norm=None) -> float:
    """optim_helper.py:186-225"""
    mean_loss, cnt = 0.0, 0
    sampler.set_epoch(epoch)
    model.eval()
    with torch.no_grad():
        for _ in range(num_loops):
            for Xs, bs, ys in dataloader:
                bs = bs.unsqueeze(1)
                Xs, bs, ys = Xs.to(rank), bs.to(rank), ys.to(rank)
                preds = model(Xs, bs)
                if grad_norm is None:
                    loss = loss_fn(preds, ys, bs)
                else:
                    losses = loss_fn.calc_loss_terms(predicts=preds, targets=ys, masks=bs)
                    loss = grad_norm.calc_total_weighted_loss_for_test(list(losses))
                mean_loss += loss * Xs.shape[0]
                cnt += Xs.shape[0]
    mean_loss /= cnt
    dist.all_reduce(mean_loss, op=dist.ReduceOp.SUM)
    return mean_loss.item() / world_size

"""Train / validation / evaluation loops behind the reference's signatures (pytorch/src/optim_helper.py:22-225:
``train``, ``test``, ``evaluate``, ``train_ddp``, ``test_ddp``).

All five are thin drivers over three pieces: ``_device_batches`` (the sample tuple of the DataLoader -> device tensors,
mask with its channel dimension), ``_train_step`` (forward, loss or GradNorm terms, backward, optional gradient
all-reduce, optimizer) and ``_eval_loss``.  ``optimizer`` is any object with ``zero_grad()`` / ``step()`` (``FlatAdam``,
a torch optimizer, or the pair train_model.py builds for GradNorm); in the DDP variants gradient averaging is the engine's
``GradAllReducer`` (``reducer``) instead of a ``DistributedDataParallel`` wrapper, and the epoch loss stays on the device
until the one all-reduce at the end."""
import typing
from logging import getLogger

import torch
import torch.distributed as dist
from torch import nn

logger = getLogger()


class AverageMeter:
    """weighted running mean with the attributes the reference's meter exposes (pytorch/src/utils.py:52-67)"""

    def __init__(self):
        self.reset()

    def reset(self) -> None:
        self.val, self.sum, self.count = 0, 0, 0

    @property
    def avg(self):
        return self.sum / self.count if self.count else 0

    def update(self, val, n: int = 1) -> None:
        self.val = val
        self.sum += val * n
        self.count += n


def _device_batches(dataloader, device, num_loops: int = 1):
    """(Xs, bs[with channel dim], ys) on `device`, `num_loops` passes over the loader"""
    for _ in range(num_loops):
        for Xs, bs, ys in dataloader:
            yield Xs.to(device), bs.unsqueeze(1).to(device), ys.to(device)


class LazyGraphedStep:
    """Engine extension (no reference counterpart): the training step as a hipGraph replay (src/graph.py) inside the
    reference's loops.  Captured on the first batch; a batch of another shape (the last, partial one of an epoch) runs
    eagerly.  Needs ``FlatAdam(capturable=True)``, no GradNorm; ``reducer`` = the data-parallel gradient averaging that
    belongs to the step (every rank then captures, replays and falls back in lockstep: same shapes on every rank)."""

    def __init__(self, model, loss_fn, optimizer, reducer=None, comm=None):
        self.model, self.loss_fn, self.optimizer, self.graphed = model, loss_fn, optimizer, None
        self.reducer, self.comm = reducer, comm
        self.failed = False     # the capture was invalidated once: every step runs eagerly from then on

    @staticmethod
    def _is_capture_invalidation(e: BaseException) -> bool:
        """a capture-unsafe call somewhere (another library, another thread) -- as opposed to an engine error, an
        out-of-memory condition or a sticky HIP error, which must not be downgraded to a warning"""
        if isinstance(e, torch.cuda.OutOfMemoryError):
            return False
        msg = str(e).lower()
        return isinstance(e, RuntimeError) and ("captur" in msg or "graph" in msg) and "out of memory" not in msg

    def _capture(self, Xs, bs, ys):
        from .graph import GraphedTrainStep
        opt = self.optimizer
        state = [t for t in (opt.flat_param, opt.exp_avg, opt.exp_avg_sq, getattr(opt, "_step_dev", None)) if t is not None]
        keep = [t.clone() for t in state]
        ok = False
        try:
            self.graphed = GraphedTrainStep(self.model, self.loss_fn, opt, Xs, bs, ys, reducer=self.reducer, comm=self.comm)
            ok = True
            logger.info(f"training step captured into a hipGraph (batch {tuple(Xs.shape)}, gradient averaging: "
                        f"{self.graphed.comm or 'none'})")
        except Exception as e:
            if not self._is_capture_invalidation(e):
                raise
            logger.warning(f"hipGraph capture of the training step failed ({type(e).__name__}: {e}); "
                           "continuing with eager steps")
            self.failed = True
        finally:
            if not ok:      # the warm-up / aborted capture must leave no trace in the optimizer state
                if Xs.is_cuda:
                    try:
                        torch.cuda.synchronize()
                    except Exception:       # noqa: BLE001  (a sticky error: the original exception is the one to report)
                        pass
                with torch.no_grad():
                    for dst, src in zip(state, keep):
                        dst.copy_(src)

    def __call__(self, Xs, bs, ys):
        if self.graphed is None and not self.failed:
            self._capture(Xs, bs, ys)
        g = self.graphed
        if g is not None and Xs.shape == g.x.shape and bs.shape == g.b.shape and ys.shape == g.y.shape:
            return g(Xs, bs, ys).clone()        # (the graph's loss buffer is overwritten by the next replay)
        return _train_step(self.model, self.loss_fn, self.optimizer, Xs, bs, ys, reducer=self.reducer).detach()


def _last_params(model):
    return (model.module if hasattr(model, "module") else model).get_last_params()


def _train_step(model, loss_fn, optimizer, Xs, bs, ys, grad_norm=None, reducer=None):
    preds = model(Xs, bs)
    if grad_norm is None:
        loss = loss_fn(preds, ys, bs)
        optimizer.zero_grad()
        loss.backward()
    else:   # GradNorm differentiates the individual terms (optim_helper.py:44-60, 167-178)
        terms = loss_fn.calc_loss_terms(predicts=preds, targets=ys, masks=bs)
        optimizer.zero_grad()
        loss = grad_norm.backward(loss_list=list(terms), last_shared_params=_last_params(model))
    if reducer is not None:
        optimizer.grad_scale = reducer.finish()
    optimizer.step()
    if grad_norm is not None:
        grad_norm.renormalize_weights()
    return loss


_step = _train_step     # name used by tests / earlier rounds


def _eval_loss(model, loss_fn, Xs, bs, ys, grad_norm=None):
    preds = model(Xs, bs)
    if grad_norm is None:
        return loss_fn(preds, ys, bs)
    terms = loss_fn.calc_loss_terms(predicts=preds, targets=ys, masks=bs)
    return grad_norm.calc_total_weighted_loss_for_test(list(terms))


def train(dataloader, model: nn.Module, loss_fn, optimizer, device: str, num_loops: int = 1,
          hide_progress_bar: bool = True, grad_norm=None, graph_step=None) -> float:
    """optim_helper.py:22-66 (`graph_step`: a LazyGraphedStep to replay the step as a hipGraph)"""
    meter = AverageMeter()
    model.train()
    for Xs, bs, ys in _device_batches(dataloader, device, num_loops):
        loss = graph_step(Xs, bs, ys) if graph_step is not None else _train_step(model, loss_fn, optimizer, Xs, bs, ys, grad_norm)
        meter.update(loss.item(), n=len(Xs))
    logger.info(f"Train error: avg loss = {meter.avg:.8f}")
    return meter.avg


def test(dataloader, model: nn.Module, loss_fn, device: str, num_loops: int = 1, hide_progress_bar: bool = True,
         grad_norm=None) -> float:
    """optim_helper.py:69-108"""
    meter = AverageMeter()
    model.eval()
    with torch.no_grad():
        for Xs, bs, ys in _device_batches(dataloader, device, num_loops):
            meter.update(_eval_loss(model, loss_fn, Xs, bs, ys, grad_norm).item(), n=len(Xs))
    logger.info(f"Valid error: avg loss = {meter.avg:.8f}")
    return meter.avg


def evaluate(*, dataloader, model: nn.Module, loss_fns: typing.Dict[str, typing.Callable], device: str,
             hide_progress_bar: bool = True) -> dict:
    """optim_helper.py:111-134: {name: AverageMeter} of every metric over the loader.  The engine's fused metrics all
    come out of ONE pass per batch: it is launched here with the union of the scales the metrics depend on, and each
    module then reads its entry."""
    from .. import ops
    from .loss_maker import _FusedMetric, merged_metric_scales
    meters = {name: AverageMeter() for name in loss_fns}
    fused = [fn for fn in loss_fns.values() if isinstance(fn, _FusedMetric)]
    scales = merged_metric_scales(fused)
    with torch.no_grad():
        for Xs, bs, ys in _device_batches(dataloader, device):
            preds = model(Xs, bs)
            if fused:
                ops.eval_metrics(preds, ys, bs, scales, fused[0].delta_meter, 0)
            values = {name: fn(preds, ys, bs) for name, fn in loss_fns.items()}
            for name, v in values.items():     # host syncs only after everything is enqueued
                meters[name].update(v.item(), n=len(Xs))
    return meters


def _epoch_mean_over_ranks(total: torch.Tensor, count: int, world_size: int) -> float:
    """sample-weighted mean of this rank, then the mean over ranks (one all-reduce per epoch)"""
    mean = total / count
    dist.all_reduce(mean, op=dist.ReduceOp.SUM)
    return mean.item() / world_size


def train_ddp(dataloader, sampler, model: nn.Module, loss_fn, optimizer, epoch: int, rank: int, world_size: int,
              num_loops: int, grad_norm=None, reducer=None, graph_step=None) -> float:
    """optim_helper.py:137-183 (`graph_step`: single-GPU runs may replay the step as a hipGraph, see LazyGraphedStep)"""
    sampler.set_epoch(epoch)
    model.train()
    total, count = 0.0, 0
    for Xs, bs, ys in _device_batches(dataloader, rank, num_loops):
        if graph_step is not None:
            loss = graph_step(Xs, bs, ys)
        else:
            loss = _train_step(model, loss_fn, optimizer, Xs, bs, ys, grad_norm, reducer)
        total = total + loss.detach() * len(Xs)
        count += len(Xs)
    return _epoch_mean_over_ranks(total, count, world_size)


def test_ddp(dataloader, sampler, model: nn.Module, loss_fn, epoch: int, rank: int, world_size: int, num_loops: int,
             grad_norm=None) -> float:
    """optim_helper.py:186-225"""
    sampler.set_epoch(epoch)
    model.eval()
    total, count = 0.0, 0
    with torch.no_grad():
        for Xs, bs, ys in _device_batches(dataloader, rank, num_loops):
            total = total + _eval_loss(model, loss_fn, Xs, bs, ys, grad_norm) * len(Xs)
            count += len(Xs)
    return _epoch_mean_over_ranks(total, count, world_size)

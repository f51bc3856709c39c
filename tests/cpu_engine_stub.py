"""TEST INFRASTRUCTURE ONLY -- a stand-in for the parts of the engine that need a GPU, so that the multi-process CONTROL
FLOW around the hot path (bench.py's measure(), script/train_model.py's per-rank function: process group, parameter
broadcast, step closure, bucketed gradient all-reduce + reducer.finish() ordering, the MAX / SUM all-reduces, barriers,
checkpoint) can be rehearsed on two CPU ranks over gloo (tests/test_dist_paths_gloo.py).  Nothing in the product imports
this file.  The model is a few ATen ops with the engine's forward signature; the optimizer is a CPU twin of FlatAdam on
the REAL flatten_parameters layout; GradAllReducer and GradNorm are the REAL classes."""
import torch
import torch.nn.functional as F
from torch import nn

import sr3d_amd
from sr3d_amd.src.optim import flatten_parameters

GradAllReducer = sr3d_amd.GradAllReducer      # the real one: host logic, backend-agnostic
LAST_OPT = None                               # the most recent FlatAdam twin (tests read the trained parameters)


class TinySR(nn.Module):
    """forward(x, b) -> (B, 4, Z, Y, X) like UNetSR; `last` is the shared last layer GradNorm differentiates"""

    def __init__(self, scale: int):
        super().__init__()
        self.scale = scale
        self.body = nn.Conv3d(5, 6, 3, padding=1)
        self.last = nn.Conv3d(6, 4, 3, padding=1)

    def get_last_params(self):
        return list(self.last.parameters())

    def forward(self, x, b):
        x0 = torch.cat([F.interpolate(x, scale_factor=self.scale, mode="nearest"), b], 1)
        return self.last(F.leaky_relu(self.body(x0), 0.01))


def make_model(config: dict) -> nn.Module:
    return TinySR(2 ** int(config["model"]["num_x2upsample"]))


class _Loss(nn.Module):
    def calc_loss_terms(self, predicts, targets, masks):
        d = predicts - targets
        return (d ** 2).mean(), (d[..., 1:] - d[..., :-1]).pow(2).mean(), (d * masks).abs().mean()

    def forward(self, predicts, targets, masks):
        return (predicts - targets).abs().mean()


def make_loss(config: dict) -> nn.Module:
    return _Loss()


class FlatAdam:
    """CPU twin of src/optim.py:FlatAdam (same layout, same attributes the callers touch)"""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        global LAST_OPT
        self.lr, self.betas, self.eps, self.capturable = float(lr), betas, float(eps), capturable
        self.grad_scale = 1.0
        self.params, self.offsets, self.flat_param, self.flat_grad = flatten_parameters(
            [p for p in params if p.requires_grad])
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat_param), torch.zeros_like(self.flat_param)
        self._host_step = 0
        LAST_OPT = self

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    @torch.no_grad()
    def step(self):
        self._host_step += 1
        g = self.flat_grad * self.grad_scale
        b1, b2 = self.betas
        self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        c1, c2 = 1 - b1 ** self._host_step, 1 - b2 ** self._host_step
        self.flat_param.addcdiv_(self.exp_avg / c1, (self.exp_avg_sq / c2).sqrt() + self.eps, value=-self.lr)


class _NoProfile:
    """bench.py's `L` argument: per-kernel HIP-event records do not exist on a CPU rank"""

    class lib:
        @staticmethod
        def sr3d_profile_read(kid, ms, work, n):
            return 0

    @staticmethod
    def check(rc, what):
        assert rc == 0, what

    @staticmethod
    def profile_enable(on):
        pass


L = _NoProfile

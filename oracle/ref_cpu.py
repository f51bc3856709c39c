"""CPU oracle for the voxel-SR training hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional restatement (stock ``torch`` CPU ops,
fp32 or fp64) of the algorithm on the reference's hot path.  It is *not* part
of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.
The shipped engine (``3d-sr-micrometeorology_amd``) never imports this module
and raises if the HIP library is missing.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference
(``/root/reference/pytorch``) in the build container and writes golden vectors
to ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
below against those vectors.  The reference itself ships no tests / golden
vectors for this path, so the vectors generated here are
the pin.

Every function cites the reference lines it restates (paths relative to
``/root/reference``).  Tensor layout everywhere: NCDHW = (B, C, z, y, x).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]


# --------------------------------------------------------------------------
# weight / bias gradient of a 3x3x3, pad-1, stride-1 convolution written out tap by tap
# --------------------------------------------------------------------------
def conv3d_weight_grad_by_taps(x: Tensor, dpre: Tensor, dtype=torch.float64) -> Tuple[Tensor, Tensor]:
    """(dW, db) of ``pre = conv3d(x, W, b, stride=1, padding=1)`` given dL/dpre -- what autograd's
    ``convolution_backward`` returns for the weight and bias of the reference's layers (nn.Conv3d at
    pytorch/model/unet.py:100,196,240; custom_conv.py:289-294), restated as the 27 plain contractions

        dW[n, c, kz, ky, kx] = sum_{b, z, y, x} dpre[b, n, z, y, x] * x[b, c, z + kz - 1, y + ky - 1, x + kx - 1]

    (out-of-range x = 0), each ONE matrix product over the voxels, accumulated in ``dtype``.  Device-agnostic: the
    full-size tests (tests/test_gpu_fullsize.py) evaluate it in fp64 on the tensors of a whole-model step where they
    live (8 M voxels x 64..194 channels is minutes of CPU time and seconds on the GPU's fp64 GEMM); it is pinned to
    autograd's result on the CPU by tests/test_oracle_golden.py."""
    B, C, Z, Y, X = x.shape
    N = dpre.shape[1]
    assert dpre.shape[0] == B and tuple(dpre.shape[2:]) == (Z, Y, X)
    dw = torch.zeros(N, C, 3, 3, 3, dtype=dtype, device=x.device)
    for kz in range(3):
        z0, z1 = max(0, 1 - kz), min(Z, Z + 1 - kz)          # output planes whose tap kz falls inside the input
        for ky in range(3):
            y0, y1 = max(0, 1 - ky), min(Y, Y + 1 - ky)
            for kx in range(3):
                x0, x1 = max(0, 1 - kx), min(X, X + 1 - kx)
                for b in range(B):     # one matrix product per z plane (a batch): parallel work for any backend
                    d = dpre[b, :, z0:z1, y0:y1, x0:x1].permute(1, 0, 2, 3).reshape(z1 - z0, N, -1).to(dtype)
                    xs = x[b, :, z0 + kz - 1:z1 + kz - 1, y0 + ky - 1:y1 + ky - 1, x0 + kx - 1:x1 + kx - 1]
                    xs = xs.permute(1, 0, 2, 3).reshape(z1 - z0, C, -1).to(dtype)
                    dw[:, :, kz, ky, kx] += torch.bmm(d, xs.transpose(1, 2)).sum(0)
    db = dpre.to(dtype).sum(dim=(0, 2, 3, 4))
    return dw, db


# --------------------------------------------------------------------------
# index permutations  (pytorch/model/voxel_shuffle.py:5-42)
# --------------------------------------------------------------------------
def unshuffle_voxels(x: Tensor, factor: int = 2) -> Tensor:
    """channel -> space. out[b,c,f*z+fz,f*y+fy,f*x+fx] = in[b,((fz*f+fy)*f+fx)*C+c,z,y,x]

    Restates pytorch/model/voxel_shuffle.py:26-42 with explicit indexing
    instead of reshape/permute.
    """
    B, C8, Z, Y, X = x.shape
    f = factor
    C = C8 // (f ** 3)
    out = x.new_empty(B, C, f * Z, f * Y, f * X)
    for fz in range(f):
        for fy in range(f):
            for fx in range(f):
                blk = ((fz * f + fy) * f + fx) * C
                out[:, :, fz::f, fy::f, fx::f] = x[:, blk:blk + C]
    return out


def shuffle_voxels(x: Tensor, factor: int = 2) -> Tensor:
    """space -> channel, inverse of :func:`unshuffle_voxels`
    (pytorch/model/voxel_shuffle.py:5-23)."""
    B, C, Z, Y, X = x.shape
    f = factor
    out = x.new_empty(B, C * f ** 3, Z // f, Y // f, X // f)
    for fz in range(f):
        for fy in range(f):
            for fx in range(f):
                blk = ((fz * f + fy) * f + fx) * C
                out[:, blk:blk + C] = x[:, :, fz::f, fy::f, fx::f]
    return out


# --------------------------------------------------------------------------
# convolution building blocks  (pytorch/model/custom_conv.py:77-126, 237-306)
# --------------------------------------------------------------------------
class KinkRecorder(list):
    """pass as ``kinks`` to RECORD the oracle's own activation decisions (in the format the forced mode consumes).

    ``audit=True`` also keeps, per activation and in the same layout as the decision masks, the pre-activation values
    (``.pre``) and the magnitude of the dot product behind each of them, ``sum_k |w_k| |x_k| + |bias|`` (``.scale``):
    what a test needs to show that a decision another fp32 evaluation took differently sat within rounding distance of
    the kink (tests/test_gpu_default_width.py)."""

    def __init__(self, audit: bool = False):
        super().__init__()
        self.audit = bool(audit)
        self.pre: List[Tensor] = []
        self.scale: List[Tensor] = []


def _audit_scale(kinks, x: Tensor, w: Tensor, b: Optional[Tensor], stride: int, unshuffle: bool = False) -> None:
    if isinstance(kinks, KinkRecorder) and kinks.audit:
        s = F.conv3d(x.detach().abs(), w.detach().abs(), None if b is None else b.detach().abs(), stride=stride, padding=1)
        kinks.scale.append(unshuffle_voxels(s, 2) if unshuffle else s)


def _act(name: Optional[str], x: Tensor, kinks=None) -> Tensor:
    """activation; with ``kinks`` (an iterator of boolean masks, one per activation in forward order) the branch of
    every element is FORCED to the given decision instead of being taken from the sign of x.  Test-only: two correct
    fp32 evaluations disagree about the branch of pre-activations within rounding distance of 0, and at default
    widths a handful of such disagreements moves parameter gradients by up to 1e-3; forcing the HIP path's decisions
    into the oracle makes both sides the same smooth function, which can then be compared at 1e-5."""
    if name is None:
        return x
    if isinstance(kinks, KinkRecorder):
        kinks.append(x.detach() > 0)
        if kinks.audit:
            kinks.pre.append(x.detach())
    elif kinks is not None:
        m = next(kinks).to(device=x.device, dtype=x.dtype)
        assert m.shape == x.shape, (m.shape, x.shape)
        slope = {"relu": 0.0, "lrelu": 0.01}[name]
        return x * (m + slope * (1 - m))
    if name == "relu":
        return F.relu(x)
    if name == "lrelu":  # nn.LeakyReLU() default slope 0.01
        return F.leaky_relu(x, 0.01)
    raise NotImplementedError(name)


def conv_with_act(sd: StateDict, prefix: str, x: Tensor, stride: int,
                  conv_mode: Optional[str], act: Optional[str], kinks=None) -> Tensor:
    """``MyConvWithAct2.forward`` (pytorch/model/custom_conv.py:111-126).

    ``conv_mode is None``: act(conv3d(x)).  Gated modes
    (``g_conv`` custom_conv.py:237-272, ``g_conv_with_separated_bias``
    :275-306): sigmoid(conv3d(x; Wg, bg)) * act(conv3d(x; Wf[, bf])).
    The bias tensors simply are or are not in the state dict.
    """
    if conv_mode is None:
        y = F.conv3d(x, sd[prefix + ".conv.weight"], sd.get(prefix + ".conv.bias"),
                     stride=stride, padding=1)
        if act is not None:
            _audit_scale(kinks, x, sd[prefix + ".conv.weight"], sd.get(prefix + ".conv.bias"), stride)
        return _act(act, y, kinks)
    if conv_mode in ("g_conv", "g_conv_with_separated_bias"):
        feat = F.conv3d(x, sd[prefix + ".conv.conv3d.weight"],
                        sd.get(prefix + ".conv.conv3d.bias"), stride=stride, padding=1)
        gate = F.conv3d(x, sd[prefix + ".conv.mask_conv3d.weight"],
                        sd.get(prefix + ".conv.mask_conv3d.bias"), stride=stride, padding=1)
        if act is not None:
            _audit_scale(kinks, x, sd[prefix + ".conv.conv3d.weight"], sd.get(prefix + ".conv.conv3d.bias"), stride)
        return torch.sigmoid(gate) * _act(act, feat, kinks)
    raise NotImplementedError(f"{conv_mode} is not supported.")


def down_block(sd: StateDict, prefix: str, x: Tensor, conv_mode: Optional[str],
               n_layers: int, kinks=None) -> Tensor:
    """``DownBlock`` (pytorch/model/unet.py:13-55): stride-2 conv then
    ``n_layers-1`` stride-1 convs, all with ReLU."""
    y = conv_with_act(sd, f"{prefix}.convs.0", x, 2, conv_mode, "relu", kinks)
    for i in range(1, n_layers):
        y = conv_with_act(sd, f"{prefix}.convs.{i}", y, 1, conv_mode, "relu", kinks)
    return y


def up_block(sd: StateDict, prefix: str, x1: Tensor, x2: Tensor,
             conv_mode: Optional[str], n_layers: int, kinks=None) -> Tensor:
    """``UpBlock`` (pytorch/model/unet.py:58-115): x3 = unshuffle(lrelu(conv(x1)+b));
    y = cat[x2, x3]; n_layers x lrelu(conv(y))."""
    u = F.conv3d(x1, sd[f"{prefix}.up.0.weight"], sd[f"{prefix}.up.0.bias"], padding=1)
    if kinks is None:
        x3 = unshuffle_voxels(F.leaky_relu(u, 0.01), 2)
    elif isinstance(kinks, KinkRecorder):
        x3 = unshuffle_voxels(F.leaky_relu(u, 0.01), 2)
        kinks.append(x3.detach() > 0)
        if kinks.audit:
            kinks.pre.append(unshuffle_voxels(u.detach(), 2))
            _audit_scale(kinks, x1, sd[f"{prefix}.up.0.weight"], sd[f"{prefix}.up.0.bias"], 1, unshuffle=True)
    else:   # the decision mask is given in the unshuffled layout (that is what the fused kernel writes)
        x3 = unshuffle_voxels(_act("lrelu", u, iter([shuffle_voxels(next(kinks).float(), 2)])), 2)
    y = torch.cat([x2, x3], dim=1)
    for i in range(n_layers):
        y = conv_with_act(sd, f"{prefix}.convs.{i}", y, 1, conv_mode, "lrelu", kinks)
    return y


def avg_pool_mask(b: Tensor) -> Tensor:
    """``nn.AvgPool3d(2, 2)`` on the building mask (pytorch/model/unet.py:156)."""
    return F.avg_pool3d(b, kernel_size=2, stride=2)


def unet_forward(sd: StateDict, model_cfg: dict, x: Tensor, b: Tensor, kinks=None) -> Tensor:
    """``UNetSR.forward`` (pytorch/model/unet.py:253-297), functional form.

    ``model_cfg`` is the ``model:`` section of the reference YAML
    (pytorch/config/default.yml:44-59).
    """
    s = 2 ** int(model_cfg["num_x2upsample"])
    nlb = int(model_cfg["n_layers_in_block"])
    cm0 = model_cfg.get("conv_mode_feat_extraction")
    cmd = model_cfg.get("conv_mode_down_block")
    cmu = model_cfg.get("conv_mode_up_block")
    f4 = model_cfg.get("num_feat4")
    has4 = f4 is not None and f4 > 0

    # nearest upsample (unet.py:143,254): x0[z,y,x] = x[z//s, y//s, x//s]
    x0 = x.repeat_interleave(s, 2).repeat_interleave(s, 3).repeat_interleave(s, 4)
    x0 = torch.cat([x0, b], 1)
    forced = kinks is not None and not isinstance(kinks, KinkRecorder)
    if forced:
        kinks = iter(kinks)
    y0 = torch.cat([conv_with_act(sd, "conv0", x0, 1, cm0, None, kinks), b], 1)

    b1 = avg_pool_mask(b)
    y1 = torch.cat([down_block(sd, "down1", y0, cmd, nlb, kinks), b1], 1)
    b2 = avg_pool_mask(b1)
    y2 = torch.cat([down_block(sd, "down2", y1, cmd, nlb, kinks), b2], 1)
    b3 = avg_pool_mask(b2)
    y3 = torch.cat([down_block(sd, "down3", y2, cmd, nlb, kinks), b3], 1)

    def latent(t: Tensor) -> Tensor:
        # unet.py:192-199: Conv3d(bias=False) + LeakyReLU, num_latent_layers times
        for i in range(int(model_cfg["num_latent_layers"])):
            _audit_scale(kinks, t, sd[f"latent_layers.{2 * i}.weight"], None, 1)
            t = _act("lrelu", F.conv3d(t, sd[f"latent_layers.{2 * i}.weight"], None, padding=1), kinks)
        return t

    if not has4:
        y = latent(y3)
    else:
        b4 = avg_pool_mask(b3)
        y4 = torch.cat([down_block(sd, "down4", y3, cmd, nlb, kinks), b4], 1)
        y = torch.cat([latent(y4), b4], 1)
        y = up_block(sd, "up4", y, y3, cmu, nlb, kinks)

    y = torch.cat([y, b3], 1)
    y = up_block(sd, "up3", y, y2, cmu, nlb, kinks)
    y = torch.cat([y, b2], 1)
    y = up_block(sd, "up2", y, y1, cmu, nlb, kinks)
    y = torch.cat([y, b1], 1)
    y = up_block(sd, "up1", y, y0, cmu, nlb, kinks)
    y = torch.cat([y, x0], 1)
    out = F.conv3d(y, sd["last.weight"], sd["last.bias"], padding=1)
    if forced:
        assert next(kinks, None) is None, "more activation decisions were recorded than the model has activations"
    return out


# --------------------------------------------------------------------------
# finite differences  (pytorch/src/math_helper.py:6-105)
# --------------------------------------------------------------------------
def central_diff(f: Tensor, axis: str, delta: float = 1.0, padding: int = 1) -> Tensor:
    """Central difference (f[i+1]-f[i-1])/(2 delta) along ``axis`` in {x,y,z}.

    The reference builds a depthwise 3x3x3 kernel with two non-zero taps
    (math_helper.py:6-60), i.e. each term is multiplied by fl(1/(2 delta)) and
    the two products are added; ``padding=0`` keeps the interior only in *all
    three* dims, ``padding=1`` zero-pads.  Its scalar triple-loop form is
    math_helper.py:63-105.
    """
    w = 1.0 / (2.0 * delta)
    dim = {"z": 2, "y": 3, "x": 4}[axis]
    if padding == 1:
        g = F.pad(f, (1, 1, 1, 1, 1, 1))
    elif padding == 0:
        g = f
    else:
        raise NotImplementedError
    n = g.shape[dim]
    hi = g.narrow(dim, 2, n - 2)
    lo = g.narrow(dim, 0, n - 2)
    d = hi * w + lo * (-w)
    # crop the two other dims to the interior
    for od in (2, 3, 4):
        if od != dim:
            d = d.narrow(od, 1, d.shape[od] - 2)
    return d


# --------------------------------------------------------------------------
# losses  (pytorch/src/loss_maker.py)
# --------------------------------------------------------------------------
def near_wall_mask(b: Tensor) -> Tensor:
    """``calc_mask_near_build_wall`` (loss_maker.py:57-83), one filter pass:
    near = 1[(box3(1-b) with zero padding > 0) * b > 0]."""
    inside = 1 - b
    box = F.avg_pool3d(F.pad(inside, (1, 1, 1, 1, 1, 1)), 3, stride=1) * 27.0
    filt = (box > 0).to(b.dtype)
    return ((filt * b) > 0).to(b.dtype)


def divergence(v: Tensor, delta: float, padding: int) -> Tensor:
    """``_calc_residual_continuity_eq`` (loss_maker.py:115-130): du/dx+dv/dy+dw/dz
    for a 3-channel (u,v,w) tensor."""
    assert v.shape[1] == 3
    return (central_diff(v[:, 0:1], "x", delta, padding)
            + central_diff(v[:, 1:2], "y", delta, padding)
            + central_diff(v[:, 2:3], "z", delta, padding))


def l1_loss(p: Tensor, t: Tensor, b: Tensor = None) -> Tensor:
    """``MyL1Loss`` (loss_maker.py:194-202): mean |p-t|, mask ignored."""
    return (p - t).abs().mean()


def l2_loss(p: Tensor, t: Tensor, b: Tensor = None) -> Tensor:
    """``MyL2Loss`` (loss_maker.py:205-213)."""
    return ((p - t) ** 2).mean()


def mixed_div_grad_terms(p: Tensor, t: Tensor, b: Tensor, w_g: float, w_d: float,
                         scales: Sequence[float], delta_meter: float = 5.0):
    """``MixedDivergenceGradientL2Loss.calc_loss_terms`` (loss_maker.py:387-437).

    Returns (mse, grd_mse, div_mse); a skipped term is the python float 0.0
    exactly as in the reference (:399-413)."""
    d = p - t
    mse = (d ** 2).mean()
    near = near_wall_mask(b)
    m = b[:, :, 1:-1, 1:-1, 1:-1] * (1 - near[:, :, 1:-1, 1:-1, 1:-1])

    grd = 0.0
    if w_g != 0.0:
        gsum = (central_diff(d, "x", 1.0, 0) ** 2 + central_diff(d, "y", 1.0, 0) ** 2
                + central_diff(d, "z", 1.0, 0) ** 2)
        grd = (gsum * m).sum() / (4 * m.sum() + 1)

    div = 0.0
    if w_d == 0.0:
        return mse, grd, div
    sc = torch.tensor(list(scales), dtype=torch.float32).to(p.device)[None, :, None, None, None]
    # NB: like the reference, the scales tensor is float32 even when p is float64
    # (torch.tensor(list_of_floats) default dtype), type promotion does the rest.
    mean_scale = sum(scales) / len(scales)  # np.mean(scales), loss_maker.py:375
    div_t = divergence(sc * t[:, 1:], delta_meter, 0)
    div_p = divergence(sc * p[:, 1:], delta_meter, 0)
    dd = (div_t - div_p) * delta_meter / mean_scale
    div = ((dd ** 2) * m).sum() / (m.sum() + 1)
    return mse, grd, div


def mixed_div_grad_loss(p, t, b, w_g, w_d, scales, delta_meter=5.0) -> Tensor:
    """``MixedDivergenceGradientL2Loss.forward`` (loss_maker.py:439-450)."""
    mse, grd, div = mixed_div_grad_terms(p, t, b, w_g, w_d, scales, delta_meter)
    return mse + w_g * grd + w_d * div


def make_loss(config: dict):
    """``make_loss`` (loss_maker.py:19-54) for the losses on the training path."""
    name = config["train"]["loss"]["name"]
    if name == "L1":
        return l1_loss
    if name == "L2":
        return l2_loss
    if name == "MixedDivergenceGradientL2Loss":
        lc = config["train"]["loss"]
        w_g = lc.get("weight_gradient_loss", 0.0)
        w_d = lc.get("weight_divergence_loss", 0.0)
        scales = list(config["data"]["stds"][1:])
        return lambda p, t, b: mixed_div_grad_loss(p, t, b, w_g, w_d, scales)
    raise NotImplementedError(f"{name} is not supported.")


# --------------------------------------------------------------------------
# PartialConv3d (dead code in the reference; custom_conv.py:129-234)
# --------------------------------------------------------------------------
def partial_conv3d(x: Tensor, mask: Optional[Tensor], weight: Tensor, bias: Optional[Tensor],
                   stride: int = 1, padding: int = 1, multi_channel: bool = True) -> Tuple[Tensor, Tensor]:
    """``PartialConv3d.forward`` with ``return_mask=True`` (custom_conv.py:176-234).  ``mask`` None = the all-ones
    mask the reference builds itself (:188-199); single-channel masks use the (1, 1, 3, 3, 3) updater."""
    if multi_channel:
        ones = torch.ones_like(weight)
        given = mask if mask is not None else torch.ones_like(x)
    else:
        ones = torch.ones(1, 1, *weight.shape[2:], dtype=weight.dtype)
        given = mask if mask is not None else torch.ones(1, 1, *x.shape[2:], dtype=x.dtype)
    upd = F.conv3d(given, ones, None, stride=stride, padding=padding)
    win = ones.shape[1] * ones.shape[2] * ones.shape[3] * ones.shape[4]
    ratio = win / (upd + 1e-8)
    upd = upd.clamp(0, 1)
    ratio = ratio * upd
    raw = F.conv3d(x * mask if mask is not None else x, weight, bias, stride=stride, padding=padding)
    if bias is not None:
        bv = bias.view(1, -1, 1, 1, 1)
        out = ((raw - bv) * ratio + bv) * upd
    else:
        out = raw * ratio
    return out, upd


# --------------------------------------------------------------------------
# parameter init + training step  (custom_conv.py:289-299, optim_helper.py:22-66,
# script/train_model.py:183)
# --------------------------------------------------------------------------
def param_shapes(model_cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    """state_dict names and shapes of ``UNetSR`` in registration order
    (pytorch/model/unet.py:119-246)."""
    ci, co = int(model_cfg["in_channels"]), int(model_cfg["out_channels"])
    f = [int(model_cfg[f"num_feat{i}"]) for i in range(4)]
    f4 = model_cfg.get("num_feat4")
    has4 = f4 is not None and f4 > 0
    nlb = int(model_cfg["n_layers_in_block"])
    cm0 = model_cfg.get("conv_mode_feat_extraction")
    cmd = model_cfg.get("conv_mode_down_block")
    cmu = model_cfg.get("conv_mode_up_block")
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def conv_act(prefix, cin, cout, mode, bias):
        if mode is None:
            out.append((f"{prefix}.conv.weight", (cout, cin, 3, 3, 3)))
            if bias:
                out.append((f"{prefix}.conv.bias", (cout,)))
        else:
            out.append((f"{prefix}.conv.conv3d.weight", (cout, cin, 3, 3, 3)))
            if bias:
                out.append((f"{prefix}.conv.conv3d.bias", (cout,)))
            out.append((f"{prefix}.conv.mask_conv3d.weight", (cout, cin, 3, 3, 3)))
            if bias or mode == "g_conv_with_separated_bias":
                out.append((f"{prefix}.conv.mask_conv3d.bias", (cout,)))

    conv_act("conv0", ci + 1, f[0], cm0, bool(model_cfg["bias_feat_extraction"]))
    widths = [f[0], f[1], f[2], f[3]] + ([int(f4)] if has4 else [])
    for lvl in range(1, len(widths)):
        conv_act(f"down{lvl}.convs.0", widths[lvl - 1] + 1, widths[lvl], cmd, False)
        for i in range(1, nlb):
            conv_act(f"down{lvl}.convs.{i}", widths[lvl], widths[lvl], cmd, False)
    for i in range(int(model_cfg["num_latent_layers"])):
        out.append((f"latent_layers.{2 * i}.weight", (f[3], f[3] + (1 if i == 0 else 0), 3, 3, 3)))

    def up(prefix, in1, in2, cout):
        conv_act(f"{prefix}.convs.0", in1 + in2, cout, cmu, False)
        for i in range(1, nlb):
            conv_act(f"{prefix}.convs.{i}", cout, cout, cmu, False)
        out.append((f"{prefix}.up.0.weight", (in1 * 8, in1, 3, 3, 3)))
        out.append((f"{prefix}.up.0.bias", (in1 * 8,)))

    if has4:
        up("up4", int(f4) + 1, f[3] + 1, f[3])
    up("up3", f[3] + 1, f[2] + 1, f[2])
    up("up2", f[2] + 1, f[1] + 1, f[1])
    up("up1", f[1] + 1, f[0] + 1, f[0])
    out.append(("last.weight", (co, f[0] + ci + 1, 3, 3, 3)))
    out.append(("last.bias", (co,)))
    return out


def random_state_dict(model_cfg: dict, seed: int = 0, dtype=torch.float32) -> StateDict:
    """Random weights with the right shapes/scales (NOT seed-for-seed identical
    to ``UNetSR.__init__``; parity tests load the reference's own weights from
    the golden files instead)."""
    g = torch.Generator().manual_seed(seed)
    sd: StateDict = {}
    for name, shape in param_shapes(model_cfg):
        if name.endswith("weight"):
            fan_in = shape[1] * 27
            sd[name] = (torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)).to(dtype)
        else:
            sd[name] = ((torch.rand(shape, generator=g) - 0.5) * 0.2).to(dtype)
    return sd


class AdamState:
    """Plain Adam (``torch.optim.Adam`` defaults used at
    pytorch/script/train_model.py:183: betas (0.9, 0.999), eps 1e-8, no weight
    decay, no amsgrad), written out so the HIP fused Adam has an explicit
    formula to match:  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
    p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""

    def __init__(self, params: StateDict, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    @torch.no_grad()
    def step(self, params: StateDict, grads: StateDict) -> None:
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        for k, p in params.items():
            g = grads[k]
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def loss_and_grads(sd: StateDict, config: dict, x: Tensor, b: Tensor, t: Tensor, kinks=None, l1_sign=None):
    """forward + loss + backward of one batch (the body of
    pytorch/src/optim_helper.py:42-47 without the optimizer step).

    ``kinks`` / ``l1_sign`` (test-only, see :func:`_act`): forced activation decisions, and for the L1 loss the forced
    sign(p - t) field.  Returns (pred, loss, dL/dpred, {name: grad})."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    pred = unet_forward(leaves, config["model"], x, b, kinks)
    pred.retain_grad()
    if l1_sign is not None and config["train"]["loss"]["name"] == "L1":
        loss = ((pred - t) * l1_sign.to(pred.dtype)).mean()
    else:
        loss = make_loss(config)(pred, t, b)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return pred.detach(), loss.detach(), pred.grad.detach(), grads


def train_step(sd: StateDict, opt: AdamState, config: dict, x: Tensor, b: Tensor, t: Tensor) -> float:
    """One ``train`` iteration (optim_helper.py:38-47): forward, loss,
    zero_grad, backward, Adam step.  ``b`` already has its channel dim."""
    _, loss, _, grads = loss_and_grads(sd, config, x, b, t)
    opt.step(sd, grads)
    return float(loss)


# --------------------------------------------------------------------------
# evaluation metrics of the final test pass  (pytorch/src/loss_maker.py:84-191, 522-745;
# called at pytorch/script/train_model.py:366-390)
# --------------------------------------------------------------------------
def vorticity(v: Tensor, delta: float, padding: int) -> Tensor:
    """``_calc_vorticity_vector`` (loss_maker.py:163-191): curl of a 3-channel (u, v, w) field,
    (dw/dy - dv/dz, du/dz - dw/dx, dv/dx - du/dy)."""
    assert v.shape[1] == 3
    gx, gy, gz = (central_diff(v, a, delta, padding) for a in ("x", "y", "z"))
    return torch.cat([gy[:, 2:3] - gz[:, 1:2], gz[:, 0:1] - gx[:, 2:3], gx[:, 1:2] - gy[:, 0:1]], dim=1)


def _interior_masked(field_fn, b: Tensor, q: Tensor, scales: Sequence[float], delta: float):
    """shared body of ``calc_residual_continuity_eq`` / ``calc_vorticity_vector`` (loss_maker.py:84-113, 133-160):
    field of the scaled velocity on the interior, zeroed inside buildings and next to walls; the voxel count is
    sum(b) - sum(near) over the interior."""
    sc = torch.tensor(list(scales), dtype=torch.float32)[None, :, None, None, None]   # fp32 like the reference
    f = field_fn(sc * q[:, 1:], delta, 0)
    near = near_wall_mask(b)
    bi, ni = b[..., 1:-1, 1:-1, 1:-1], near[..., 1:-1, 1:-1, 1:-1]
    f = f * bi
    f = f * (1 - ni)
    return f, bi.sum() - ni.sum()


def eval_metrics(p: Tensor, t: Tensor, b: Tensor, stds: Sequence[float], delta: float = 5.0, lev: int = 0,
                 eps: float = 1e-30) -> Dict[str, Tensor]:
    """Every metric ``script/train_model.py:366-379`` evaluates (plus the L2 variants and the target-side residual of
    ``ResidualContinuity.calc_both_pred_and_target``), keyed like ``_lib.EVAL_INDEX`` of the engine."""
    d = p - t
    ad, sq = d.abs(), d ** 2
    m4 = torch.broadcast_to(b, ad.shape)
    near = near_wall_mask(b)
    n4 = torch.broadcast_to(near, ad.shape)
    out = {"L1": ad.mean(), "L2": sq.mean(),
           "MaskedL1": (m4 * ad).sum() / (m4.sum() + eps), "MaskedL2": (m4 * sq).sum() / (m4.sum() + eps),
           "MaskedL1NearWall": (n4 * ad).sum() / (n4.sum() + eps), "MaskedL2NearWall": (n4 * sq).sum() / (n4.sum() + eps)}
    rp, n1 = _interior_masked(divergence, b, p, stds[1:], delta)
    rt, _ = _interior_masked(divergence, b, t, stds[1:], delta)
    out["ResidualContinuity"] = rp.abs().sum() / n1
    out["ResidualContinuityTarget"] = rt.abs().sum() / n1
    out["AbsDiffDivergence"] = (rp - rt).abs().sum() / n1
    op, _ = _interior_masked(vorticity, b, p, stds[1:], delta)
    ot, _ = _interior_masked(vorticity, b, t, stds[1:], delta)
    out["DiffOmegaNorm"] = torch.linalg.norm(op - ot, dim=1, keepdim=True).sum() / n1
    sc = torch.tensor(list(stds[1:]), dtype=torch.float32)[None, :, None, None, None]
    dv = torch.linalg.norm(p[:, 1:] * sc - t[:, 1:] * sc, dim=1, keepdim=True)
    dT = (p[:, 0:1] - t[:, 0:1]).abs() * stds[0]
    out["AbsDiffTemperature"] = (b * dT).sum() / (b.sum() + eps)
    out["DiffVelocityNorm"] = (b * dv).sum() / (b.sum() + eps)
    bl = b[:, :, lev]
    out["AbsDiffTemperatureLev"] = (bl * dT[:, :, lev]).sum() / (bl.sum() + eps)
    out["DiffVelocityNormLev"] = (bl * dv[:, :, lev]).sum() / (bl.sum() + eps)
    return out


def weighted_lp_loss(p: Tensor, t: Tensor, b: Tensor, weight: float, power: int) -> Tensor:
    """``WeightedL1Loss`` / ``WeightedL2Loss`` (loss_maker.py:216-255)."""
    e = (p - t).abs() if power == 1 else (p - t) ** 2
    m = torch.broadcast_to(b, e.shape)
    inside = (m * e).sum() / (m.sum() + 1)
    outside = ((1 - m) * e).sum() / ((1 - m).sum() + 1)
    return (weight * inside + outside) / (weight + 1)


def mixed_gradient_l2_loss(p: Tensor, t: Tensor, b: Tensor, w_g) -> Tensor:
    """``MixedGradientL2Loss`` (loss_maker.py:258-301)."""
    mse, grd, _ = mixed_div_grad_terms(p, t, b, 0.0 if not w_g else float(w_g), 0.0, [1.0, 1.0, 1.0])
    return mse if not w_g else mse + w_g * grd


def mixed_gradient_weighted_l2_terms(p: Tensor, t: Tensor, b: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """``MixedGradientWeightedL2Loss.calc_loss_terms`` (loss_maker.py:313-343): squared error averaged over the fluid voxels
    (mask 1) and over the buildings (mask 0), each with its ``+ 1`` in the denominator and the mask broadcast over the 4
    channels, and the gradient term of the mixed losses (interior, off the walls; the 4-channel mask sums to 4 * sum(M))."""
    d = p - t
    sq = d ** 2
    m = torch.broadcast_to(b, sq.shape)
    one = (m * sq).sum() / (m.sum() + 1)
    zero = ((1 - m) * sq).sum() / ((1 - m).sum() + 1)
    near = near_wall_mask(b)
    gm = (m * (1 - torch.broadcast_to(near, sq.shape)))[:, :, 1:-1, 1:-1, 1:-1]
    g2 = sum(central_diff(d, ax, 1.0, padding=0) ** 2 for ax in ("x", "y", "z"))
    return one, zero, (g2 * gm).sum() / (gm.sum() + 1)


def mixed_gradient_weighted_l2_loss(p: Tensor, t: Tensor, b: Tensor, weight: float, w_g: float) -> Tensor:
    """``MixedGradientWeightedL2Loss.forward`` (loss_maker.py:345-355)"""
    one, zero, grd = mixed_gradient_weighted_l2_terms(p, t, b)
    return (weight * one + zero) / (weight + 1) + w_g * grd


def channelwise_mse(p: Tensor, t: Tensor, i_channel: int) -> Tensor:
    """``ChannelwiseMse`` (loss_maker.py:753-764)"""
    return ((p[:, i_channel] - t[:, i_channel]) ** 2).mean()


def ssim3d(img1: Tensor, img2: Tensor, mask: Tensor, window_size: int = 11, sigma: float = 1.5, max_val: float = 1.0,
           eps: float = 1e-7, use_gaussian: bool = True, size_average: bool = True) -> Tensor:
    """``_ssim_3D`` (src/ssim.py:52-115) with its dense depthwise w x w x w window; mask already of img1's shape."""
    if use_gaussian:
        w = torch.tensor([math.exp(-((i - window_size // 2) ** 2) / float(2 * sigma ** 2)) for i in range(window_size)])
    else:
        w = torch.ones(window_size)
    w = (w / w.sum()).unsqueeze(1)
    w3 = w.mm(w.mm(w.t()).reshape(1, -1)).reshape(window_size, window_size, window_size).float()
    C = img1.shape[1]
    win = w3[None, None].expand(C, 1, -1, -1, -1).contiguous().to(img1.dtype)

    def filt(t):
        return F.conv3d(t, win, padding=window_size // 2, groups=C)
    a, b = img1 * mask, img2 * mask
    wt = filt(mask) + eps
    mu1, mu2 = filt(a) / wt, filt(b) / wt
    s1 = filt(a * a) / wt - mu1.pow(2)
    s2 = filt(b * b) / wt - mu2.pow(2)
    s12 = filt(a * b) / wt - mu1 * mu2
    c1, c2 = (max_val * 0.01) ** 2, (max_val * 0.03) ** 2
    smap = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1.pow(2) + mu2.pow(2) + c1) * (s1 + s2 + c2))
    return smap.mean() if size_average else smap

"""Convolution wrappers with the reference's module/parameter names
(pytorch/model/custom_conv.py), executing through the fused HIP ops.

``nn.Conv3d`` instances are used purely as *parameter containers*: they give
the same state_dict keys, the same default initialisation and the same RNG
consumption order as the reference (custom_conv.py:289-299), but their
``forward`` is never called -- the arithmetic is ``ops.Conv3dAct`` /
``ops.GatedConv3dAct``.  Inputs may be a tensor or a list of tensors; a list
is a *virtual* channel concat (the reference's ``torch.cat``)."""
import typing

import torch
from torch import nn

from .. import ops

TensorOrList = typing.Union[torch.Tensor, typing.Sequence[torch.Tensor]]


def _as_list(x: TensorOrList) -> typing.List[torch.Tensor]:
    return [x] if isinstance(x, torch.Tensor) else list(x)


def _act_name(act: typing.Optional[nn.Module]) -> typing.Optional[str]:
    if act is None:
        return None
    if isinstance(act, nn.ReLU):
        return "relu"
    if isinstance(act, nn.LeakyReLU):
        if abs(act.negative_slope - 0.01) > 1e-12:
            raise NotImplementedError("the fused kernels implement LeakyReLU with the default slope 0.01")
        return "lrelu"
    raise NotImplementedError(f"activation {act} is not supported by the fused kernels")


def _check_3x3x3(kernel_size, stride, padding, dilation, groups):
    ks = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
    if ks != 3 or padding != 1 or dilation != 1 or groups != 1 or stride not in (1, 2):
        raise NotImplementedError("the MI355X engine implements the model's 3x3x3, pad 1, stride 1|2 convolutions")


class GatedConv3d(nn.Module):
    """custom_conv.py:237-272: feature and gate branch share the ``bias`` flag."""

    separated_bias = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        _check_3x3x3(kernel_size, stride, padding, dilation, groups)
        self.stride = stride
        self.conv3d = nn.Conv3d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.mask_conv3d = nn.Conv3d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                     True if self.separated_bias else bias)
        self.sigmoid = nn.Sigmoid()
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight)

    def gated_forward(self, srcs, act: typing.Optional[str], dual: bool = False):
        return ops.gated_conv3d_act(srcs, self.conv3d.weight, self.mask_conv3d.weight, self.conv3d.bias,
                                    self.mask_conv3d.bias, act=act, stride=self.stride, dual=dual)


class GatedConv3dWithSeparatedBias(GatedConv3d):
    """custom_conv.py:275-306: the gate always has a bias."""

    separated_bias = True


class MyConvWithAct2(nn.Module):
    """custom_conv.py:77-126."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 conv_mode=None, act=None):
        super().__init__()
        self.act = act
        self.conv_mode = conv_mode
        self._act_name = _act_name(act)
        self.stride = stride
        if conv_mode is None:
            _check_3x3x3(kernel_size, stride, padding, dilation, groups)
            self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        elif conv_mode == "g_conv":
            self.conv = GatedConv3d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        elif conv_mode == "g_conv_with_separated_bias":
            self.conv = GatedConv3dWithSeparatedBias(in_channels, out_channels, kernel_size, stride, padding, dilation,
                                                     groups, bias)
        else:
            raise NotImplementedError(f"{conv_mode} is not supported.")

    def forward(self, input: TensorOrList, defer_act_bwd: bool = False, dual: bool = False):
        """``defer_act_bwd`` (engine extension, plain LeakyReLU layers): the caller guarantees that the output feeds exactly one
        engine convolution, whose input-gradient kernel then applies this layer's activation backward (ops.Conv3dAct).
        ``dual`` (engine extension): return ``(y, y)`` -- two handles of one tensor for an output with two consumers (the
        U-Net's skip tensors); a gated layer then adds the two incoming gradients inside its activation backward
        (ops.GatedConv3dAct), any other layer returns the same tensor twice and autograd adds as usual."""
        srcs = _as_list(input)
        if self.conv_mode is None:
            y = ops.conv3d_act(srcs, self.conv.weight, self.conv.bias, act=self._act_name, stride=self.stride,
                               defer_act_bwd=defer_act_bwd)
            return (y, y) if dual else y
        return self.conv.gated_forward(srcs, self._act_name, dual=dual)


class PartialConv3d(nn.Conv3d):
    """custom_conv.py:129-234 (NVIDIA partial convolution, 3-D): conv(x * mask) renormalised by the number of valid
    inputs under each window, plus the updated mask.  Not reachable from ``UNetSR`` (no conv_mode selects it); kept as
    an op on the engine's kernels: mask product, the fused convolution, mask statistics and renormalisation are HIP,
    the bias is folded exactly as in the reference."""

    def __init__(self, *args, **kwargs):
        self.multi_channel = kwargs.pop("multi_channel", False)
        self.return_mask = kwargs.pop("return_mask", False)
        super().__init__(*args, **kwargs)
        _check_3x3x3(self.kernel_size, self.stride[0], self.padding[0], self.dilation[0], self.groups)
        # weight_maskUpdater is all ones: (out, in, 3,3,3) or (1, 1, 3,3,3); only its window size matters
        self.slide_winsize = (self.in_channels if self.multi_channel else 1) * 27
        self.last_size = (None, None, None, None, None)
        self.update_mask = None
        self.mask_ratio = None

    def forward(self, input, mask_in=None):
        assert len(input.shape) == 5
        if mask_in is not None or self.last_size != tuple(input.shape):
            self.last_size = tuple(input.shape)
            with torch.no_grad():
                if mask_in is None:
                    shape = tuple(input.shape) if self.multi_channel else (1, 1) + tuple(input.shape[2:])
                    mask = torch.ones(shape, dtype=input.dtype, device=input.device)
                else:
                    mask = mask_in
                self.update_mask, self.mask_ratio = ops.pconv_mask_update(mask, self.stride[0], self.slide_winsize)
        x = ops.MulMask.apply(input, mask_in) if mask_in is not None else input
        raw = ops.conv3d_act([x], self.weight, self.bias, act=None, stride=self.stride[0])
        output = ops.PconvScale.apply(raw, self.bias, self.update_mask, self.mask_ratio)
        if self.return_mask:
            # the reference's updated mask has one (identical) channel per output channel when multi_channel
            um = self.update_mask
            return output, (um.expand(-1, self.out_channels, -1, -1, -1) if self.multi_channel else um)
        return output

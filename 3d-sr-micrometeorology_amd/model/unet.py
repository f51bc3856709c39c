"""UNetSR with the reference's constructor, parameter names and forward
signature (pytorch/model/unet.py:13-297), running on the fused HIP ops.

Differences in *execution* (not in results):
 - no ``torch.cat``: every block receives its inputs as a list (virtual concat);
 - ``nn.Upsample`` + the first concat are one small kernel (``ops.upsample_cat``);
 - ``UpBlock.up`` (conv + bias + LeakyReLU + VoxelUnshuffle) is one kernel whose
   epilogue scatters straight into the unshuffled layout."""
import typing
from logging import getLogger

import torch
from torch import nn

from .. import ops
from .custom_conv import MyConvWithAct2, _as_list
from .voxel_shuffle import VoxelUnshuffle

logger = getLogger()


class DownBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, bias: bool, conv_mode: str, n_layers_in_block: int):
        super().__init__()
        assert n_layers_in_block >= 1
        layers = [MyConvWithAct2(in_channels, out_channels, kernel_size=3, stride=2, padding=1, bias=bias,
                                 conv_mode=conv_mode, act=nn.ReLU())]
        for _ in range(n_layers_in_block - 1):
            layers.append(MyConvWithAct2(out_channels, out_channels, kernel_size=3, padding=1, bias=bias,
                                         conv_mode=conv_mode, act=nn.ReLU()))
        self.convs = nn.Sequential(*layers)

    def forward(self, x, dual: bool = False):
        """``dual``: the block's output as two handles (for the next block / for the skip connection; MyConvWithAct2.forward)"""
        y = x
        for i, layer in enumerate(self.convs):
            y = layer(y, dual=dual and i + 1 == len(self.convs))
        return y


class UpBlock(nn.Module):
    def __init__(self, in1_channels: int, in2_channels: int, out_channels: int, bias: bool, conv_mode: str,
                 n_layers_in_block: int):
        super().__init__()
        assert n_layers_in_block >= 1
        layers = [MyConvWithAct2(in1_channels + in2_channels, out_channels, kernel_size=3, padding=1, bias=bias,
                                 conv_mode=conv_mode, act=nn.LeakyReLU())]
        for _ in range(n_layers_in_block - 1):
            layers.append(MyConvWithAct2(out_channels, out_channels, kernel_size=3, padding=1, bias=bias,
                                         conv_mode=conv_mode, act=nn.LeakyReLU()))
        self.convs = nn.Sequential(*layers)
        # parameter container + markers; executed as ONE fused kernel in forward()
        self.up = nn.Sequential(nn.Conv3d(in1_channels, in1_channels * 8, kernel_size=3, padding=1), nn.LeakyReLU(),
                                VoxelUnshuffle(factor=2))

    def forward(self, x1, x2, defer_act_bwd: bool = False) -> torch.Tensor:
        """``defer_act_bwd``: set by UNetSR, where every LeakyReLU output of this block feeds exactly one convolution"""
        x3 = ops.conv3d_act(_as_list(x1), self.up[0].weight, self.up[0].bias, act="lrelu", stride=1, unshuffle=True,
                            defer_act_bwd=defer_act_bwd)
        y = _as_list(x2) + [x3]
        for layer in self.convs:
            y = layer(y, defer_act_bwd=defer_act_bwd)
        return y


class UNetSR(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, num_feat0: int, num_feat1: int, num_feat2: int,
                 num_feat3: int, num_feat4: int, num_x2upsample: int, num_latent_layers: int,
                 bias_feat_extraction: bool, conv_mode_feat_extraction: str, conv_mode_down_block: str,
                 conv_mode_up_block: str, n_layers_in_block: int, **kwargs):
        super().__init__()
        # Engine extension (absent from the reference's YAML = fp32, the reference's only precision):
        # `model: {storage_dtype: bf16}` keeps every activation and activation gradient INSIDE the network in bfloat16
        # (BASELINE configs[4]): bf16 MFMA with fp32 accumulation, half the HBM bytes and half the saved-for-backward
        # memory.  Parameters, their gradients and the optimizer state stay fp32 (master weights); the network's
        # inputs and its prediction are fp32 at the boundary, so losses, metrics and loaders are unchanged.
        sd = kwargs.get("storage_dtype", None)
        if sd in (None, "fp32", "float32"):
            self.act_dtype = torch.float32
        elif sd in ("bf16", "bfloat16"):
            self.act_dtype = torch.bfloat16
        else:
            raise NotImplementedError(f"storage_dtype {sd!r}: the engine stores activations as fp32 or bf16")
        logger.info(f"conv_mode_feat_extraction = {conv_mode_feat_extraction}")
        logger.info(f"conv_mode_down_block = {conv_mode_down_block}")
        logger.info(f"conv_mode_up_block = {conv_mode_up_block}")

        self.scale = 2 ** num_x2upsample
        self.up0 = nn.Upsample(scale_factor=self.scale, mode="nearest")  # marker; fused into ops.upsample_cat
        self.conv0 = MyConvWithAct2(in_channels + 1, num_feat0, kernel_size=3, padding=1, bias=bias_feat_extraction,
                                    conv_mode=conv_mode_feat_extraction, act=None)
        self.down = nn.AvgPool3d(kernel_size=2, stride=2)  # marker; executed by ops.avgpool2

        def down_block(cin, cout):
            return DownBlock(in_channels=cin + 1, out_channels=cout, bias=False, conv_mode=conv_mode_down_block,
                             n_layers_in_block=n_layers_in_block)

        self.down1 = down_block(num_feat0, num_feat1)
        self.down2 = down_block(num_feat1, num_feat2)
        self.down3 = down_block(num_feat2, num_feat3)
        self.down4 = None
        has4 = num_feat4 is not None and num_feat4 > 0
        if has4:
            self.down4 = down_block(num_feat3, num_feat4)

        latent_layers = []
        for i in range(num_latent_layers):
            _in = num_feat3 if i > 0 else num_feat3 + 1
            latent_layers.append(nn.Conv3d(_in, num_feat3, kernel_size=3, padding=1, bias=False))
            latent_layers.append(nn.LeakyReLU())
        self.latent_layers = nn.Sequential(*latent_layers)

        def up_block(c1, c2, cout):
            return UpBlock(in1_channels=c1 + 1, in2_channels=c2 + 1, out_channels=cout, bias=False,
                           conv_mode=conv_mode_up_block, n_layers_in_block=n_layers_in_block)

        self.up4 = up_block(num_feat4, num_feat3, num_feat3) if has4 else None
        self.up3 = up_block(num_feat3, num_feat2, num_feat2)
        self.up2 = up_block(num_feat2, num_feat1, num_feat1)
        self.up1 = up_block(num_feat1, num_feat0, num_feat0)
        self.last = nn.Conv3d(num_feat0 + in_channels + 1, out_channels, kernel_size=3, padding=1, bias=True)

    def get_last_params(self) -> typing.List[torch.nn.Parameter]:
        return list(self.last.parameters())

    def _latent(self, srcs) -> torch.Tensor:
        y = srcs
        for layer in self.latent_layers:
            if isinstance(layer, nn.Conv3d):
                y = ops.conv3d_act(_as_list(y), layer.weight, layer.bias, act="lrelu", stride=1, defer_act_bwd=True)
        return y

    def forward(self, x: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        x0 = ops.upsample_cat(x, b, self.scale)  # cat[up0(x), b]; inputs carry no gradient
        b = b.detach().contiguous()
        # the mask pyramid is a function of the mask alone: computed in fp32, each level rounded ONCE if stored as bf16
        b1 = ops.avgpool2(b)
        b2 = ops.avgpool2(b1)
        b3 = ops.avgpool2(b2)
        b4 = ops.avgpool2(b3) if not (self.down4 is None and self.up4 is None) else None
        if self.act_dtype != torch.float32:
            x0, b, b1, b2, b3 = (t.to(self.act_dtype) for t in (x0, b, b1, b2, b3))
            b4 = b4.to(self.act_dtype) if b4 is not None else None
        # f0 .. f3 have two consumers each, the next block and the skip connection: two handles of one tensor, so that the
        # producing gated layer adds the two gradients inside its activation backward (no elementwise add pass in between)
        dual = ops.FUSE_SKIP_GRAD_ADD
        f0, f0s = self.conv0([x0], dual=True) if dual else (self.conv0([x0]),) * 2
        f1, f1s = self.down1([f0, b], dual=True) if dual else (self.down1([f0, b]),) * 2
        f2, f2s = self.down2([f1, b1], dual=True) if dual else (self.down2([f1, b1]),) * 2
        has4 = not (self.down4 is None and self.up4 is None)
        f3, f3s = self.down3([f2, b2], dual=True) if (dual and has4) else (self.down3([f2, b2]),) * 2

        if not has4:
            y = self._latent([f3, b3])
        else:
            f4 = self.down4([f3, b3])
            y = self._latent([f4, b4])
            y = self.up4([y, b4], [f3s, b3], defer_act_bwd=True)
        # (every LeakyReLU output below has ONE consumer, an engine convolution: its input-gradient epilogue applies the
        #  activation backward -- SURVEY K9 -- where the kernel has that epilogue; ops.Conv3dAct falls back otherwise)
        y = self.up3([y, b3], [f2s, b2], defer_act_bwd=True)
        y = self.up2([y, b2], [f1s, b1], defer_act_bwd=True)
        y = self.up1([y, b1], [f0s, b], defer_act_bwd=True)
        w, bias = self.last.weight, self.last.bias
        # (bf16 storage: the prediction leaves `last` as fp32, the accumulator's value -- not rounded to bf16 and cast back)
        return ops.conv3d_act([y, x0], w, bias, act=None, stride=1, out_fp32=self.act_dtype != torch.float32)

"""Voxel shuffle helpers (interface of reference pytorch/model/voxel_shuffle.py).

On the training path the unshuffle is never run as an op: it is the scatter
epilogue of the up-convolution kernel (``ops.Conv3dAct(unshuffle=True)``).  The
functions here are host-side index permutations kept for API completeness
(the reference's ``VoxelShuffle`` is not used by its model either)."""
import torch
from torch import nn


def unshuffle_voxels(x: torch.Tensor, factor: int) -> torch.Tensor:
    """out[b,c,f z+fz,f y+fy,f x+fx] = in[b,((fz f+fy) f+fx) C + c,z,y,x]  (voxel_shuffle.py:26-42)"""
    b, c8, d, h, w = x.shape
    f = factor
    c = c8 // f ** 3
    return x.view(b, f, f, f, c, d, h, w).permute(0, 4, 5, 1, 6, 2, 7, 3).reshape(b, c, f * d, f * h, f * w)


def shuffle_voxels(x: torch.Tensor, factor: int) -> torch.Tensor:
    """inverse of :func:`unshuffle_voxels` (voxel_shuffle.py:5-23)"""
    b, c, d, h, w = x.shape
    f = factor
    y = x.view(b, c, d // f, f, h // f, f, w // f, f).permute(0, 3, 5, 7, 1, 2, 4, 6)
    return y.reshape(b, c * f ** 3, d // f, h // f, w // f)


class VoxelUnshuffle(nn.Module):
    def __init__(self, factor: int):
        super().__init__()
        self.factor = factor

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return unshuffle_voxels(x, self.factor)


class VoxelShuffle(nn.Module):
    def __init__(self, factor: int):
        super().__init__()
        self.factor = factor

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return shuffle_voxels(x, self.factor)

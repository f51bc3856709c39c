"""SSIM3D with the reference's interface (pytorch/src/ssim.py:17-189): ``gaussian`` / ``uniform`` / ``create_window_3D``,
``ssim3D(img1, img2, mask, ...)`` and the ``SSIM3D`` module.  The arithmetic is the engine's separable three-pass
kernel (``ops.ssim3d``): the reference's w x w x w window is the triple outer product of its 1-D taps, so filtering
along x, y and z with those taps gives the same sums with 33 instead of 1331 taps per voxel.  Forward only."""
from logging import getLogger
import torch

from .. import ops

logger = getLogger()


def gaussian(window_size: int, sigma: float) -> torch.Tensor:
    """normalised 1-D Gaussian taps centred on window_size // 2 (fp32, as the reference's window)"""
    offs = torch.arange(window_size, dtype=torch.float64) - (window_size // 2)
    taps = torch.exp(-offs.square() / (2.0 * float(sigma) ** 2)).to(torch.float32)
    return taps / taps.sum()


def uniform(window_size: int) -> torch.Tensor:
    return torch.ones(window_size) / window_size


def create_window_3D(window_size: int, channel: int, sigma: float, use_gaussian: bool = True) -> torch.Tensor:
    """the dense (channel, 1, w, w, w) window the reference convolves with (kept for API compatibility; the
    engine itself only needs the 1-D taps)"""
    w = gaussian(window_size, sigma) if use_gaussian else uniform(window_size)
    w3 = torch.einsum("i,j,k->ijk", w, w, w).float()
    return w3[None, None].expand(channel, 1, window_size, window_size, window_size).contiguous()


def _taps(window_size: int, sigma: float, use_gaussian: bool):
    return (gaussian(window_size, sigma) if use_gaussian else uniform(window_size)).tolist()


def ssim3D(img1, img2, mask, window_size=11, sigma=1.5, size_average=True, max_val=1.0, use_gaussian=True):
    """ssim.py:118-143 (its call of ``_ssim_3D`` leaves eps at that function's default 1e-7)"""
    return ops.ssim3d(img1, img2, mask, _taps(window_size, sigma, use_gaussian), max_val, 1e-7, size_average)


class SSIM3D(torch.nn.Module):
    def __init__(self, window_size=11, sigma=1.5, size_average=True, max_val=1.0, eps=1e-7, use_gaussian=True):
        super().__init__()
        self.window_size, self.sigma, self.size_average = window_size, sigma, size_average
        self.channel, self.max_val, self.eps, self.use_gaussian = 4, max_val, eps, use_gaussian
        self.window = create_window_3D(window_size, self.channel, sigma, use_gaussian)
        logger.info(f"Use Gaussian = {self.use_gaussian}")

    def forward(self, img1, img2, mask):
        assert img1.shape == img2.shape == mask.shape      # the reference's contract (ssim.py:63)
        return ops.ssim3d(img1, img2, mask, _taps(self.window_size, self.sigma, self.use_gaussian), self.max_val,
                          self.eps, self.size_average)

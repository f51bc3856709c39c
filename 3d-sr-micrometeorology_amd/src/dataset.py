"""Sample pipeline of the reference (pytorch/src/dataset.py:17-197) -- the step in front of the hot path.

On disk: ``*_HR.npy`` (4, z, y, x) with NaN inside buildings, ``*_LR_x04.npy`` (4, z/4, y/4, x/4) and one
``hr_is_in_build.npy``.  ``__getitem__`` returns ``(lr, building_mask, hr)`` exactly as the reference does:
normalise with (mean, std), clamp to [0, 1], random 3-D crop, NaN -> nan_value, LR re-sampled on the crop.

The reference materialises the LR volume at HR resolution (nearest) before cropping and then takes every
``scale``-th voxel of the crop; here that composition is evaluated directly as an index gather,
``lr[c, (o + s*i) // s]``, so the 52 MB intermediate per sample is never built.  The HR file is memory-mapped and
only the crop window is read and normalised (the window commutes with the element-wise steps).  Results are
identical to the reference's, bit for bit (tests/test_dataset.py)."""
import glob
import os
import pathlib
import typing
from logging import getLogger

import numpy as np
import torch
from torch.utils.data import Dataset

from .utils import RandomCrop3D

logger = getLogger()


class DatasetWithoutAligningResolution(Dataset):
    def __init__(self, data_dirs: typing.List[pathlib.Path], hr_3d_build_path: pathlib.Path,
                 means: typing.List[float] = [0.0, 0.0, 0.0, 0.0], stds: typing.List[float] = [1.0, 1.0, 1.0, 1.0],
                 nan_value: float = 0.0, scale_factor: int = 4, hr_org_size: tuple = (32, 320, 320),
                 hr_crop_size: tuple = (16, 64, 64), num_channels: int = 4, dtype: torch.dtype = torch.float32,
                 use_cropping: bool = True, use_clipping: bool = True, datasize: int = None, seed: int = 42,
                 lr_scaling: float = None, max_height_index: int = 32, max_discarded_lr_z_index: int = None,
                 raw: bool = False, **kwargs):
        self.nan_value, self.scale_factor, self.num_channels, self.dtype = nan_value, scale_factor, num_channels, dtype
        self.lr_scaling, self.max_height_index = lr_scaling, max_height_index
        self.raw = raw      # hand out un-normalised windows; the GPU does the rest (src/device_pipeline.py)
        self.max_discarded_lr_z_index = max_discarded_lr_z_index
        assert scale_factor in (4, 8), "Not implemented yet."
        assert all(c % scale_factor == 0 for c in hr_crop_size)
        self.hr_org_size = tuple(hr_org_size)
        self.use_cropping, self.use_clipping = use_cropping, use_clipping
        if max_discarded_lr_z_index is not None:
            assert max_height_index == 32 and hr_crop_size[0] == 32 and self.hr_org_size[0] == 32
        self.random_3d_crop = RandomCrop3D(self.hr_org_size, hr_crop_size)

        hr_files, lr_files = [], []
        for d in data_dirs:
            hr_files += sorted(glob.glob(str(pathlib.Path(d) / "*_HR.npy")))
            lr_files += sorted(glob.glob(str(pathlib.Path(d) / f"*_LR_x{scale_factor:02}.npy")))
        assert len(hr_files) == len(lr_files)
        if datasize is not None and datasize < len(hr_files):
            import sklearn.utils
            hr_files, lr_files = sklearn.utils.shuffle(hr_files, lr_files, random_state=seed, n_samples=datasize)
        for h, l in zip(hr_files, lr_files):
            assert os.path.basename(h).split("_")[0] == os.path.basename(l).split("_")[0]
        self.hr_files, self.lr_files = list(hr_files), list(lr_files)

        build = torch.from_numpy(np.load(str(hr_3d_build_path))).to(dtype)[0:1]
        assert not torch.isnan(build).any()
        # 1 = fluid, 0 = building (the file stores "is in building")
        self.fluid_mask = torch.where(build == 0, torch.ones_like(build), torch.zeros_like(build))
        self.means = torch.tensor(means, dtype=dtype)[:, None, None, None]
        self.stds = torch.tensor(stds, dtype=dtype)[:, None, None, None]

    def __len__(self):
        return len(self.hr_files)

    def _normalise(self, x: torch.Tensor, clip: bool) -> torch.Tensor:
        y = (x - self.means) / self.stds
        return torch.clamp(y, min=0.0, max=1.0) if clip else y

    def _raw_crops(self, idx: int):
        """the sample's HR / mask / LR windows, still in physical units.  The files are memory-mapped and only the
        window is read: the reference loads and normalises the whole 52 MB HR volume and then keeps 1/25 of it; the
        window commutes with every element-wise step, so the values are identical and the work per sample drops from
        13 M to 0.5 M elements (what lets a CPU loader keep up with a step of a few hundred ms)."""
        s = self.scale_factor
        hr_mm = np.load(self.hr_files[idx], mmap_mode="r")
        lr = torch.from_numpy(np.load(self.lr_files[idx])).to(self.dtype)     # 1/64 of the HR volume: read whole
        assert tuple(hr_mm.shape[-3:]) == tuple(v * s for v in lr.shape[-3:])
        oz_, oy_, ox_ = self.hr_org_size
        assert tuple(hr_mm.shape[-2:]) == (oy_, ox_) and hr_mm.shape[-3] >= oz_
        if self.use_cropping:
            z0, y0, x0 = self.random_3d_crop.draw()
            cz, cy, cx = self.random_3d_crop.crop_sz
        else:
            z0 = y0 = x0 = 0
            cz, cy, cx = min(oz_, self.max_height_index), oy_, ox_
        hr = torch.from_numpy(np.ascontiguousarray(hr_mm[:, z0:z0 + cz, y0:y0 + cy, x0:x0 + cx])).to(self.dtype)
        bldg = self.fluid_mask[0, z0:z0 + cz, y0:y0 + cy, x0:x0 + cx]
        # nearest-upsample, crop, then every s-th voxel  ==  gather at (o + s*i) // s
        iz = (z0 + s * torch.arange(cz // s)) // s
        iy = (y0 + s * torch.arange(cy // s)) // s
        ix = (x0 + s * torch.arange(cx // s)) // s
        return lr[:, iz][:, :, iy][:, :, :, ix], bldg, hr

    def __getitem__(self, idx: int):
        lr_c, bldg, hr_c = self._raw_crops(idx)
        if self.raw:      # normalisation happens on the GPU (src/device_pipeline.py)
            return lr_c.contiguous(), torch.nan_to_num(bldg, nan=self.nan_value).contiguous(), hr_c.contiguous()
        if self.lr_scaling is not None:
            lr_c = self.lr_scaling * lr_c
        hr_c = torch.nan_to_num(self._normalise(hr_c, self.use_clipping), nan=self.nan_value)
        lr_c = torch.nan_to_num(self._normalise(lr_c, True), nan=self.nan_value)
        bldg = torch.nan_to_num(bldg, nan=self.nan_value)
        lr_c = lr_c.squeeze()  # the reference's `.squeeze()` after F.interpolate also drops size-1 spatial dims
        if self.max_discarded_lr_z_index is not None and self.max_discarded_lr_z_index > 0:
            lr_c[:, :self.max_discarded_lr_z_index] = self.nan_value
        return lr_c.contiguous(), bldg.contiguous(), hr_c.contiguous()

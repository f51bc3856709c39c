"""Small helpers with the reference's names (pytorch/src/utils.py)."""
import os
import random
from logging import getLogger

import numpy as np
import torch

logger = getLogger()


class RandomCrop3D:
    """random (z, y, x) window; one ``torch.randint`` draw per axis whose crop is smaller than the
    volume, in z, y, x order (utils.py:14-49) -- same RNG consumption as the reference"""

    def __init__(self, img_sz, crop_sz):
        assert all(i >= c for i, c in zip(img_sz, crop_sz))
        self.img_sz, self.crop_sz = tuple(img_sz), tuple(crop_sz)

    def draw(self):
        lows = []
        for sz, c in zip(self.img_sz, self.crop_sz):
            lows.append(0 if sz == c else int(torch.randint(sz - c, (1,)).item()))
        return tuple(lows)

    def __call__(self, x):
        z, y, xx = self.draw()
        cz, cy, cx = self.crop_sz
        return x[..., z:z + cz, y:y + cy, xx:xx + cx]


def set_seeds(seed: int = 42, use_deterministic: bool = False) -> None:
    """utils.py:70-92 (the cudnn switches have no counterpart: the HIP kernels are deterministic by design)"""
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    if use_deterministic:
        torch.use_deterministic_algorithms(True, warn_only=True)


def seed_worker(worker_id: int):
    worker_seed = torch.initial_seed() % 2 ** 32
    np.random.seed(worker_seed)
    random.seed(worker_seed)


def get_torch_generator(seed: int = 42) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def count_model_params(model: torch.nn.Module) -> int:
    return sum(p.numel() for p in model.parameters() if p.requires_grad)

"""Directory discovery, chronological split and DataLoader construction of the reference
(pytorch/src/dataloader.py:60-192)."""
import glob
import math
import os
import pathlib
import typing
from logging import getLogger

from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from .dataset import DatasetWithoutAligningResolution
from .utils import get_torch_generator, seed_worker

logger = getLogger()


def get_all_new_lr_data_dir_paths(root_dir: pathlib.Path, dir_name: str = "10") -> typing.List[pathlib.Path]:
    return [pathlib.Path(p) for p in sorted(glob.glob(str(pathlib.Path(root_dir) / dir_name / "*"))) if os.path.isdir(p)]


def _ordered_split(items, test_fraction: float):
    """sklearn.model_selection.train_test_split(shuffle=False) for a float test size:
    n_test = ceil(f * n) and the train part is the complement, n_train = n - n_test (sklearn's
    _validate_shuffle_split; floor((1 - f) * n) differs from it for some (f, n), e.g. f = 0.3, n = 90)"""
    n = len(items)
    n_test = int(math.ceil(test_fraction * n))
    n_train = n - n_test
    return items[:n_train], items[n_train:]


def split_into_train_valid_test_dirs(all_data_dirs, train_valid_test_ratios):
    """chronological 60/20/20 split (dataloader.py:88-104)"""
    rest, test_dirs = _ordered_split(list(all_data_dirs), train_valid_test_ratios[-1])
    valid_fraction = train_valid_test_ratios[1] / (train_valid_test_ratios[0] + train_valid_test_ratios[1])
    train_dirs, valid_dirs = _ordered_split(rest, valid_fraction)
    return {"train": train_dirs, "valid": valid_dirs, "test": test_dirs}


def make_dataloaders(data_dirs, hr_3d_build_path, means=[0.0] * 4, stds=[1.0] * 4, nan_value: float = 0.0,
                     hr_org_size: tuple = (32, 320, 320), hr_crop_size: tuple = (16, 64, 64), rank: int = None,
                     world_size: int = None, batch_size: int = 32, num_workers: int = 2, seed: int = 0,
                     datasizes: typing.Dict[str, int] = {}, use_clipping: bool = True, lr_scaling: float = None,
                     max_discarded_lr_z_index: int = None, scale_factor: int = 4, **kwargs):
    """dataloader.py:107-192; per-rank batch = batch_size // world_size"""
    loaders, samplers = {}, {}
    for kind in ["train", "valid", "test"]:
        dataset = DatasetWithoutAligningResolution(
            data_dirs=data_dirs[kind], hr_3d_build_path=hr_3d_build_path, means=means, stds=stds, nan_value=nan_value,
            hr_org_size=hr_org_size, hr_crop_size=hr_crop_size, datasize=datasizes.get(kind, None), seed=seed,
            use_clipping=use_clipping, lr_scaling=lr_scaling, max_discarded_lr_z_index=max_discarded_lr_z_index,
            scale_factor=scale_factor)
        train = kind == "train"
        if world_size is None or rank is None:
            loaders[kind] = DataLoader(dataset, batch_size=batch_size, drop_last=train, shuffle=train, pin_memory=True,
                                       num_workers=num_workers, worker_init_fn=seed_worker,
                                       generator=get_torch_generator(seed))
        else:
            samplers[kind] = DistributedSampler(dataset, num_replicas=world_size, rank=rank, seed=seed, shuffle=train,
                                                drop_last=train)
            loaders[kind] = DataLoader(dataset, sampler=samplers[kind], batch_size=batch_size // world_size,
                                       pin_memory=True, num_workers=num_workers, worker_init_fn=seed_worker,
                                       generator=get_torch_generator(seed), drop_last=train)
        if rank in (None, 0):
            logger.info(f"{kind}: dataset size = {len(dataset)}, batch num = {len(loaders[kind])}")
    return loaders, samplers

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


def get_all_data_dir_paths(root_dir: pathlib.Path) -> typing.List[pathlib.Path]:
    """dataloader.py:20-60: the three hourly directory sets 03 / 04 / 05, one entry per date and set, ordered
    04, 03, 05 within a date (the chronological order of the simulations)"""
    root_dir = pathlib.Path(root_dir)
    per_set = {n: [p for p in sorted(glob.glob(str(root_dir / n / "*"))) if os.path.isdir(p)] for n in ("03", "04", "05")}
    assert len(per_set["03"]) == len(per_set["04"]) == len(per_set["05"])
    out = []
    for d04, d03, d05 in zip(per_set["04"], per_set["03"], per_set["05"]):
        assert os.path.basename(d04) == os.path.basename(d03) == os.path.basename(d05)   # same date
        out += [pathlib.Path(d04), pathlib.Path(d03), pathlib.Path(d05)]
    return out


def data_dirs_of_config(config: dict, data_dir_root: pathlib.Path) -> typing.List[pathlib.Path]:
    """the directory sets a config names (train_model.py:134-147, dataloader.py:203-216)"""
    names = config["data"]["data_dir_names"]
    if names == ["03", "04", "05"]:
        return get_all_data_dir_paths(data_dir_root)
    if names == ["10"]:
        return get_all_new_lr_data_dir_paths(data_dir_root)
    if names == ["20"]:
        return get_all_new_lr_data_dir_paths(data_dir_root, dir_name="20")
    raise Exception(f"Data dirs {names} are not supported.")


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
                     max_discarded_lr_z_index: int = None, scale_factor: int = 4, device_pipeline=None, **kwargs):
    """dataloader.py:107-192; per-rank batch = batch_size // world_size.

    ``device_pipeline`` (a cuda device or None): workers deliver raw windows, normalise / clamp / NaN-fill run on that
    GPU one batch ahead of the consumer (src/device_pipeline.py); the batches are bit-identical either way."""
    loaders, samplers = {}, {}
    for kind in ["train", "valid", "test"]:
        dataset = DatasetWithoutAligningResolution(
            data_dirs=data_dirs[kind], hr_3d_build_path=hr_3d_build_path, means=means, stds=stds, nan_value=nan_value,
            hr_org_size=hr_org_size, hr_crop_size=hr_crop_size, datasize=datasizes.get(kind, None), seed=seed,
            use_clipping=use_clipping, lr_scaling=lr_scaling, max_discarded_lr_z_index=max_discarded_lr_z_index,
            scale_factor=scale_factor, raw=device_pipeline is not None)
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
        if device_pipeline is not None:
            from .device_pipeline import DeviceBatchPipeline
            loaders[kind] = DeviceBatchPipeline(loaders[kind], device_pipeline, means, stds, nan_value, use_clipping,
                                                lr_scaling, max_discarded_lr_z_index)
        if rank in (None, 0):
            logger.info(f"{kind}: dataset size = {len(dataset)}, batch num = {len(loaders[kind])}")
    return loaders, samplers


def make_evaluation_dataloader_without_random_cropping(config: dict, data_dir_root: pathlib.Path, batch_size: int = 1,
                                                       num_workers: int = 2, max_height_index: int = 32):
    """dataloader.py:195-246: the test split, whole domain (no crop), HR not clipped, in file order.  Every
    ``config['data']`` key that shapes a sample is forwarded (lr_scaling, max_discarded_lr_z_index, scale_factor,
    datasizes['test']), so a model trained on LR with the lowest levels discarded is evaluated on the same kind of
    input."""
    data_dir_root = pathlib.Path(data_dir_root)
    all_data_dirs = data_dirs_of_config(config, data_dir_root)
    test_dirs = split_into_train_valid_test_dirs(all_data_dirs, config["data"]["train_valid_test_ratios"])["test"]
    d = config["data"]
    dataset = DatasetWithoutAligningResolution(
        data_dirs=test_dirs, hr_3d_build_path=all_data_dirs[0].parent / "hr_is_in_build.npy", means=d["means"],
        stds=d["stds"], nan_value=d["nan_value"], hr_org_size=tuple(d["hr_org_size"]),
        hr_crop_size=tuple(d["hr_crop_size"]), use_cropping=False, use_clipping=False,
        datasize=d["datasizes"]["test"], seed=d["seed"], lr_scaling=d.get("lr_scaling", None),
        max_height_index=max_height_index, max_discarded_lr_z_index=d.get("max_discarded_lr_z_index", None),
        scale_factor=d.get("scale_factor", 4))
    return DataLoader(dataset, batch_size=batch_size, drop_last=False, shuffle=False, pin_memory=False,
                      num_workers=num_workers)

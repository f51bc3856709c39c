"""Deterministic synthetic data tree in the reference's on-disk format
(data/DL_data/10/<date>/<time>_HR.npy, <time>_LR_x04.npy, 10/hr_is_in_build.npy)."""
import pathlib

import numpy as np

HR = (8, 16, 24)   # z, y, x  (scale 4 -> LR 2x4x6)
CASES = {
    "crop": dict(means=[300.0, -6.5, -9.1, -3.5], stds=[8.4, 14.4, 21.6, 7.0], hr_org_size=HR, hr_crop_size=(8, 8, 12)),
    "crop_z": dict(means=[300.0, 0.0, 0.0, 0.0], stds=[10.0, 20.0, 20.0, 7.0], hr_org_size=HR, hr_crop_size=(4, 16, 8),
                   use_clipping=False, lr_scaling=0.9, nan_value=-1.0),
    "full": dict(means=[300.0, -6.5, -9.1, -3.5], stds=[8.4, 14.4, 21.6, 7.0], hr_org_size=HR, hr_crop_size=(8, 8, 12),
                 use_cropping=False, use_clipping=False),
    "subset": dict(hr_org_size=HR, hr_crop_size=(8, 16, 24), datasize=5, seed=7),
}


def write_synthetic_tree(tmp, HR=HR, days=5) -> pathlib.Path:
    rng = np.random.default_rng(2024)
    root = pathlib.Path(tmp) / "DL_data"
    build = (rng.random((1,) + HR) < 0.15).astype(np.float32)
    build[:, HR[0] // 2:] = 0  # buildings only near the ground
    (root / "10").mkdir(parents=True)
    np.save(root / "10" / "hr_is_in_build.npy", build)
    for day in range(days):
        d = root / "10" / f"2013080{day + 1}"
        d.mkdir()
        for t in range(2):
            hr = rng.normal(size=(4,) + HR).astype(np.float32) * np.array([8, 14, 21, 7], np.float32)[:, None, None, None]
            hr += np.array([302, -6, -9, -3], np.float32)[:, None, None, None]
            hr[:, build[0] > 0] = np.nan
            blocks = hr.reshape(4, HR[0] // 4, 4, HR[1] // 4, 4, HR[2] // 4, 4)
            with np.errstate(all="ignore"):
                lr = np.nanmean(blocks, axis=(2, 4, 6)).astype(np.float32)   # NaN where a block is all building
            np.save(d / f"{day}{t}00_HR.npy", hr)
            np.save(d / f"{day}{t}00_LR_x04.npy", lr)
    return root


# evaluation loader (whole domain, 32 levels): a config['data'] section as in config/missing_below_43m.yml
HR32 = (32, 8, 12)
EVAL_CONFIG = {"data": {"data_dir_names": ["10"], "train_valid_test_ratios": [0.6, 0.2, 0.2],
                        "means": [300.0, -6.5, -9.1, -3.5], "stds": [8.4, 14.4, 21.6, 7.0], "nan_value": 0.0,
                        "hr_org_size": list(HR32), "hr_crop_size": list(HR32), "datasizes": {"train": None, "valid": None,
                                                                                            "test": 3},
                        "seed": 11, "lr_scaling": 0.95, "max_discarded_lr_z_index": 2, "scale_factor": 4}}

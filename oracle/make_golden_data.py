"""Golden vectors for the sample pipeline: run the REFERENCE's
DatasetWithoutAligningResolution / split on tiny synthetic .npy files (build container only).
The files themselves are regenerated from a seed by tests/data_fixture.py, only outputs are stored."""
import os
import sys
import tempfile

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/pytorch")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from data_fixture import CASES, EVAL_CONFIG, HR32, write_synthetic_tree  # noqa: E402
from src.dataloader import (get_all_new_lr_data_dir_paths, make_evaluation_dataloader_without_random_cropping,  # noqa: E402
                            split_into_train_valid_test_dirs)
from src.dataset import DatasetWithoutAligningResolution  # noqa: E402

out = {}
with tempfile.TemporaryDirectory() as tmp:
    root = write_synthetic_tree(tmp)
    dirs = get_all_new_lr_data_dir_paths(root)
    split = split_into_train_valid_test_dirs(dirs, [0.6, 0.2, 0.2])
    out["split_sizes"] = np.array([len(split[k]) for k in ("train", "valid", "test")])
    out["split_first"] = np.array([os.path.basename(str(split[k][0])) for k in ("train", "valid", "test")])
    for name, kw in CASES.items():
        ds = DatasetWithoutAligningResolution(data_dirs=dirs, hr_3d_build_path=root / "10" / "hr_is_in_build.npy", **kw)
        out[f"{name}/len"] = np.array(len(ds))
        for idx in (0, 3):
            torch.manual_seed(100 + idx)
            lr, b, hr = ds[idx]
            out[f"{name}/{idx}/lr"], out[f"{name}/{idx}/b"], out[f"{name}/{idx}/hr"] = lr.numpy(), b.numpy(), hr.numpy()
# the reference's own evaluation loader (dataloader.py:195-246) on a 32-level tree, LR levels 0-1 discarded
with tempfile.TemporaryDirectory() as tmp:
    root = write_synthetic_tree(tmp, HR=HR32, days=10)
    loader = make_evaluation_dataloader_without_random_cropping(EVAL_CONFIG, root, batch_size=1, num_workers=0)
    out["eval/len"] = np.array(len(loader))
    out["eval/files"] = np.array([os.path.basename(f) for f in loader.dataset.hr_files])
    for i, (lr, b, hr) in enumerate(loader):
        out[f"eval/{i}/lr"], out[f"eval/{i}/b"], out[f"eval/{i}/hr"] = lr.numpy(), b.numpy(), hr.numpy()
np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "dataset.npz"), **out)
print("dataset.npz", len(out), "arrays")

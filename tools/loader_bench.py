#!/usr/bin/env python3
"""Can the sample pipeline feed the MI355X step?  

Writes a synthetic data tree at the reference's real sizes (HR (4, 32, 320, 320) fp32 = 52 MB per sample, default.yml)
and times, per worker process, samples/s of
  whole  : the reference's order of operations -- load the whole HR volume, normalise and clamp all of it, upsample
           LR to HR, stack, crop, NaN-fill (restated here; pytorch/src/dataset.py:139-197)
  window : this engine's CPU path -- memory-map, cut the crop window, normalise the window (src/dataset.py)
  raw    : this engine's device pipeline -- workers only cut raw windows, the GPU normalises (src/device_pipeline.py)
against the demand of the training step: default.yml trains on batches of 32 crops of 32 x 64 x 64 = 4.19 M voxels.

usage: python tools/loader_bench.py [--files 12] [--samples 48] [--step-voxels-per-s 2.4e7]"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

HR = (32, 320, 320)
CROP = (32, 64, 64)
MEANS, STDS = [302.0, -6.5, -9.1, -3.5], [8.4, 14.4, 21.6, 7.0]


def write_tree(root, n_files):
    rng = np.random.default_rng(0)
    d = os.path.join(root, "10", "20130801")
    os.makedirs(d)
    build = (rng.random((1,) + HR) < 0.15).astype(np.float32)
    build[:, HR[0] // 2:] = 0
    np.save(os.path.join(root, "10", "hr_is_in_build.npy"), build)
    for i in range(n_files):
        hr = (rng.standard_normal((4,) + HR, dtype=np.float32) * np.array(STDS, np.float32)[:, None, None, None]
              + np.array(MEANS, np.float32)[:, None, None, None])
        hr[:, build[0] > 0] = np.nan
        lr = hr.reshape(4, HR[0] // 4, 4, HR[1] // 4, 4, HR[2] // 4, 4)[:, :, 0, :, 0, :, 0].copy()
        np.save(os.path.join(d, f"{i:04d}_HR.npy"), hr)
        np.save(os.path.join(d, f"{i:04d}_LR_x04.npy"), lr)
    return [d]


def whole_volume_sample(hr_path, lr_path, fluid, means, stds, crop_draw):
    """the reference's __getitem__ (dataset.py:139-197), restated for timing"""
    hr = torch.from_numpy(np.load(hr_path)).to(torch.float32)
    lr = torch.from_numpy(np.load(lr_path)).to(torch.float32)
    lr = F.interpolate(lr.unsqueeze(0), size=hr.shape[-3:], mode="nearest").squeeze()
    hr = torch.clamp((hr - means) / stds, 0.0, 1.0)
    lr = torch.clamp((lr - means) / stds, 0.0, 1.0)
    st = torch.cat([fluid, hr, lr], dim=0)[:, :HR[0]]
    z, y, x = crop_draw()
    st = torch.nan_to_num(st[:, z:z + CROP[0], y:y + CROP[1], x:x + CROP[2]], nan=0.0)
    lr_c = F.interpolate(st[5:].unsqueeze(0), scale_factor=0.25, mode="nearest").squeeze()
    return lr_c, st[0], st[1:5]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=12)
    ap.add_argument("--samples", type=int, default=48)
    ap.add_argument("--step-voxels-per-s", type=float, default=2.4e7, help="measured training throughput (bench.py)")
    args = ap.parse_args()
    torch.set_num_threads(1)          # one DataLoader worker = one process with one thread
    import sr3d_amd  # noqa: F401
    from sr3d_amd.src.dataset import DatasetWithoutAligningResolution
    with tempfile.TemporaryDirectory() as tmp:
        dirs = write_tree(tmp, args.files)
        build = os.path.join(tmp, "10", "hr_is_in_build.npy")
        kw = dict(data_dirs=[__import__("pathlib").Path(d) for d in dirs], hr_3d_build_path=build, means=MEANS, stds=STDS,
                  hr_org_size=HR, hr_crop_size=CROP)
        ds_win = DatasetWithoutAligningResolution(**kw)
        ds_raw = DatasetWithoutAligningResolution(raw=True, **kw)
        means = torch.tensor(MEANS)[:, None, None, None]
        stds = torch.tensor(STDS)[:, None, None, None]
        res = {}
        for name, fn in (("whole", lambda i: whole_volume_sample(ds_win.hr_files[i], ds_win.lr_files[i], ds_win.fluid_mask,
                                                                 means, stds, ds_win.random_3d_crop.draw)),
                         ("window", lambda i: ds_win[i]), ("raw", lambda i: ds_raw[i])):
            n = args.samples if name != "whole" else max(4, args.samples // 4)
            fn(0)
            t0 = time.perf_counter()
            for k in range(n):
                fn(k % args.files)
            res[name] = n / (time.perf_counter() - t0)
        vox = CROP[0] * CROP[1] * CROP[2]
        demand = args.step_voxels_per_s / vox
        print(f"training step demand: {demand:.0f} crops/s ({args.step_voxels_per_s / 1e6:.1f} M voxels/s, crop {CROP})")
        for k, v in res.items():
            print(f"{k:7s}: {v:8.1f} samples/s per worker  -> {demand / v:6.1f} workers needed "
                  f"(the reference runs 2, script/train_model.py:158)")
        # the whole-volume path and the window path must agree (same seed -> same crops)
        torch.manual_seed(0)
        a = whole_volume_sample(ds_win.hr_files[1], ds_win.lr_files[1], ds_win.fluid_mask, means, stds,
                                ds_win.random_3d_crop.draw)
        torch.manual_seed(0)
        b = ds_win[1]
        assert all(torch.equal(u, v) for u, v in zip(a, b)), "window path differs from the whole-volume path"
        print("window path == whole-volume path (bit-identical sample)")


if __name__ == "__main__":
    main()

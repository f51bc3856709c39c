#!/usr/bin/env python3
"""Forward-only inference + fused evaluation metrics at the reference's evaluation shape:
LR (1,4,8,80,80) -> HR (1,4,32,320,320), default.yml widths.  Prints ms and, for the metrics kernel, GB/s of its
algorithmic 36 B per voxel.  usage: python tools/eval_bench.py [--iters 10]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import sr3d_amd  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = "cuda:0"
    cfg = bench.make_config("l1")
    torch.manual_seed(0)
    model = sr3d_amd.make_model(cfg).to(dev).eval()
    stds = cfg["data"]["stds"]
    for B, hr in ((1, (32, 320, 320)), (4, (80, 320, 320))):
        x, b, y = bench.synthetic_batch(B, hr, 4, 5, dev)
        vox = B * hr[0] * hr[1] * hr[2]
        with torch.no_grad():
            if B == 1:
                t_inf = timed(lambda: model(x, b), args.iters)
                print(f"inference  B={B} HR {hr}: {t_inf:8.2f} ms  = {vox / t_inf / 1e3:.1f} M voxels/s")
            p = torch.rand_like(y)
            sr3d_amd.ops._eval_cache["refs"] = None

            def metrics():
                sr3d_amd.ops._eval_cache["refs"] = None    # defeat the per-batch cache: time the kernel
                return sr3d_amd.ops.eval_metrics(p, y, b, stds)
            t_m = timed(metrics, args.iters)
            print(f"metrics    B={B} HR {hr}: {t_m:8.3f} ms  = {36.0 * vox / (t_m * 1e-3) / 1e9:.0f} GB/s of 36 B/voxel "
                  f"(all {len(sr3d_amd._lib.EVAL_INDEX)} metrics in one pass; the reference runs 10 modules = 10+ passes)")


if __name__ == "__main__":
    main()

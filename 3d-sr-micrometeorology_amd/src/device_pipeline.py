"""Batches prepared ON the GPU, one batch ahead of the training step.

The reference's sample pipeline (pytorch/src/dataset.py:139-197) runs entirely in DataLoader worker processes: load
52 MB, normalise all of it, crop, NaN-fill.  At MI355X step rates (the whole 80x320x320 volume in 350 ms; a batch of
32 training crops in ~0.2 s) two such workers deliver a few samples per second and the GPU would wait.  Here the
workers only cut the raw windows out of the memory-mapped files (``DatasetWithoutAligningResolution(raw=True)``);
the rest happens on the device:

    pinned host batch --(H2D on a side stream)--> raw device batch --sr3d_preprocess (same stream)--> (Xs, bs, ys)

while the main stream is still computing the previous step.  The values are bit-identical to the CPU path
(tests/test_gpu_pipeline.py)."""
from typing import Iterable, Optional, Sequence

import torch

from .. import ops


class DeviceBatchPipeline:
    def __init__(self, loader: Iterable, device, means: Sequence[float], stds: Sequence[float], nan_value: float = 0.0,
                 use_clipping: bool = True, lr_scaling: Optional[float] = None,
                 max_discarded_lr_z_index: Optional[int] = None):
        self.loader, self.device = loader, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceBatchPipeline prepares batches on the GPU")
        self.means, self.stds = [float(v) for v in means], [float(v) for v in stds]
        self.nan_value, self.use_clipping, self.lr_scaling = float(nan_value), bool(use_clipping), lr_scaling
        self.discard = int(max_discarded_lr_z_index or 0)
        self.stream = torch.cuda.Stream(self.device)

    def __len__(self):
        return len(self.loader)

    @property
    def dataset(self):
        return self.loader.dataset

    def _stage(self, batch):
        if batch is None:
            return None
        lr_raw, bldg, hr_raw = batch
        with torch.cuda.stream(self.stream):
            lr_d = lr_raw.to(self.device, non_blocking=True)
            hr_d = hr_raw.to(self.device, non_blocking=True)
            bs = bldg.to(self.device, non_blocking=True)
            ys = ops.preprocess(hr_d, self.means, self.stds, None, self.use_clipping, self.nan_value, 0)
            Xs = ops.preprocess(lr_d, self.means, self.stds, self.lr_scaling, True, self.nan_value, self.discard)
            # the reference's per-sample `.squeeze()` after the LR resampling also drops size-1 channel / spatial dims
            Xs = Xs.reshape([Xs.shape[0]] + [d for d in Xs.shape[1:] if d != 1])
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return (Xs, bs, ys), ev

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._stage(next(it, None))
        while nxt is not None:
            (Xs, bs, ys), ev = nxt
            nxt = self._stage(next(it, None))       # H2D + preprocessing of the NEXT batch overlap this step
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in (Xs, bs, ys):                  # allocated on the side stream, consumed on the main one
                t.record_stream(cur)
            yield Xs, bs, ys

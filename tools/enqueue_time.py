#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one training step (no sync) against how long the GPU needs to run it?"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import sr3d_amd  # noqa: E402

dev = torch.device("cuda:0")
cfg = bench.make_config("l1")
torch.manual_seed(42)
model = sr3d_amd.make_model(cfg).to(dev)
loss_fn = sr3d_amd.make_loss(cfg)
opt = sr3d_amd.FlatAdam(model.parameters(), lr=1e-4)
x, b, y = bench.synthetic_batch(1, (80, 320, 320), 4, 1234, dev)


def step():
    pred = model(x, b)
    loss = loss_fn(pred, y, b)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(5):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3), tot.append((t2 - t0) * 1e3)
print(json.dumps({"enqueue_ms": enq, "step_ms": tot}))

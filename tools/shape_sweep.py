#!/usr/bin/env python3
"""fwd / dgrad / wgrad error of conv3d_act against torch CPU float64 for a list of shapes (debug aid)."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import sr3d_amd  # noqa: E402
from sr3d_amd import ops  # noqa: E402


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    shapes = [(7, 10, 36), (4, 8, 32), (4, 8, 64), (5, 6, 16)]
    chans = [(40, 72), (72, 40), (40, 40), (32, 72), (64, 32), (16, 8), (40, 64)]
    if len(sys.argv) > 1:
        chans = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for shape, (cin, cout) in itertools.product(shapes, chans):
        g = torch.Generator().manual_seed(1)
        x = (torch.rand(2, cin, *shape, generator=g, dtype=torch.float64) - 0.5).requires_grad_(True)
        w = (torch.randn(cout, cin, 3, 3, 3, generator=g, dtype=torch.float64) * 0.2).requires_grad_(True)
        ref = F.conv3d(x, w, None, padding=1)
        gy = torch.rand(ref.shape, generator=g, dtype=torch.float64) - 0.5
        ref.backward(gy)
        xd = x.detach().float().cuda().requires_grad_(True)
        wd = w.detach().float().cuda().requires_grad_(True)
        y = ops.conv3d_act([xd], wd, None, act=None)
        y.backward(gy.float().cuda())
        print(f"{shape} cin={cin:3d} cout={cout:3d}  fwd {relerr(y, ref):.2e}  dgrad {relerr(xd.grad, x.grad):.2e}"
              f"  wgrad {relerr(wd.grad, w.grad):.2e}", flush=True)


if __name__ == "__main__":
    main()

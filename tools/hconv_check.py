#!/usr/bin/env python3
"""Split-f16 conv kernel (SR3D_SPLIT_F16=1) against the fp32 Winograd kernel and an fp64 reference on a few shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import sr3d_amd  # noqa: E402,F401
from sr3d_amd import ops  # noqa: E402


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def run(mode, fn):
    os.environ["SR3D_SPLIT_F16"] = mode
    out = fn()
    torch.cuda.synchronize()
    return out


def main():
    dev = "cuda:0"
    torch.manual_seed(0)
    cases = [  # (cs, cout, grid, kind, scale of x)
        ([64], 64, (8, 16, 64), "p", 1.0), ([64, 1, 65], 48, (6, 10, 40), "p", 1.0), ([128], 128, (4, 8, 32), "g", 1.0),
        ([33], 72, (5, 7, 33), "u", 1.0), ([64], 64, (8, 16, 64), "p", 1e-7), ([40], 130, (9, 9, 70), "p", 300.0),
    ]
    if len(sys.argv) > 1 and sys.argv[1] == "deep":   # large GEMM-K: error against fp64, forward and input gradient
        cases = [([1032], 129, (4, 8, 32), "p", 1.0), ([257], 2056, (2, 4, 32), "p", 1.0), ([514], 256, (4, 8, 32), "p", 1.0)]
        for cs, cout, (Z, Y, X), kind, sc in cases:
            x = (torch.randn(1, cs[0], Z, Y, X, device=dev)).requires_grad_(True)
            w = (torch.randn(cout, cs[0], 3, 3, 3, device=dev) * (2.0 / (27 * cs[0])) ** 0.5).requires_grad_(True)
            gy = torch.randn(1, cout, Z, Y, X, device=dev)
            x64, w64 = x.detach().double().cpu().requires_grad_(True), w.detach().double().cpu().requires_grad_(True)
            y64 = F.conv3d(x64, w64, None, padding=1)
            y64.backward(gy.double().cpu())
            for mode in ("0", "2"):
                os.environ["SR3D_SPLIT_F16"] = mode
                x.grad = None
                w.grad = None
                y = ops.conv3d_act([x], w, None, act=None, stride=1)
                y.backward(gy)
                torch.cuda.synchronize()
                d = (y.detach().cpu().double() - y64.detach())
                dg = (x.grad.cpu().double() - x64.grad)
                print(f"K={cs[0]} N={cout} mode={mode}: fwd rel {rel(y.cpu(), y64.detach()):.2e} mean-err/rms {d.mean().item() / y64.std().item():.2e}"
                      f" | dgrad rel {rel(x.grad.cpu(), x64.grad):.2e} mean-err/rms {dg.mean().item() / x64.grad.std().item():.2e}")
        os.environ["SR3D_SPLIT_F16"] = "0"
        return
    for cs, cout, (Z, Y, X), kind, sc in cases:
        srcs = [((torch.rand(2, c, Z, Y, X, device=dev) - 0.3) * sc).requires_grad_(c > 1) for c in cs]
        cin = sum(cs)
        wf = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).requires_grad_(True)
        bias = (torch.randn(cout, device=dev) * 0.1 * sc).requires_grad_(True)
        x64 = torch.cat([s.detach().double() for s in srcs], 1).cpu()
        if kind == "g":
            wg = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).requires_grad_(True)
            fn = lambda: ops.gated_conv3d_act(srcs, wf, wg, None, bias, act="relu", stride=1)  # noqa: E731
            ref = torch.sigmoid(F.conv3d(x64, wg.detach().double().cpu(), bias.detach().double().cpu(), padding=1)) * \
                torch.relu(F.conv3d(x64, wf.detach().double().cpu(), None, padding=1))
        elif kind == "u":
            fn = lambda: ops.conv3d_act(srcs, wf, bias, act="lrelu", unshuffle=True)  # noqa: E731
            ref = None
        else:
            fn = lambda: ops.conv3d_act(srcs, wf, bias, act="lrelu", stride=1)  # noqa: E731
            ref = F.leaky_relu(F.conv3d(x64, wf.detach().double().cpu(), bias.detach().double().cpu(), padding=1), 0.01)
        outs = {}
        for mode in ("0", "2"):
            for s in srcs:
                s.grad = None
            wf.grad = None
            y = run(mode, fn)
            torch.manual_seed(1)
            gy = torch.rand_like(y)
            os.environ["SR3D_SPLIT_F16"] = mode
            y.backward(gy)
            torch.cuda.synchronize()
            outs[mode] = (y.detach().clone(), [s.grad.clone() for s in srcs if s.grad is not None], wf.grad.clone())
        y0, g0, w0 = outs["0"]
        y1, g1, w1 = outs["2"]
        msg = f"{kind} cs={cs} cout={cout} grid={Z}x{Y}x{X} sc={sc:g}: fwd split-vs-wino {rel(y1, y0):.2e}"
        msg += "  dgrad " + " ".join(f"{rel(a, b):.2e}" for a, b in zip(g1, g0)) + f"  wgrad {rel(w1, w0):.1e}"
        if ref is not None:
            msg += f"  | vs fp64: wino {rel(y0.cpu(), ref):.2e} split {rel(y1.cpu(), ref):.2e}"
        print(msg)
    os.environ["SR3D_SPLIT_F16"] = "0"


if __name__ == "__main__":
    main()

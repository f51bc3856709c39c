#!/usr/bin/env python3
"""Per-layer timing of the conv kernels at the default.yml shapes (HIP events).
usage: python tools/layer_bench.py [--grid Z Y X] [--only NAME] [--iters N]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sr3d_amd  # noqa: E402
from sr3d_amd import ops  # noqa: E402

# (name, [source channel counts], cout, stride, level, kind)  kind: g=gated, p=plain, u=unshuffle
LAYERS = [
    ("conv0", [5], 64, 1, 0, "g"),
    ("down1.0", [64, 1], 128, 2, 0, "g"), ("down1.1", [128], 128, 1, 1, "g"),
    ("down2.0", [128, 1], 128, 2, 1, "g"), ("down2.1", [128], 128, 1, 2, "g"),
    ("down3.0", [128, 1], 256, 2, 2, "g"), ("down3.1", [256], 256, 1, 3, "g"),
    ("down4.0", [256, 1], 256, 2, 3, "g"), ("down4.1", [256], 256, 1, 4, "g"),
    ("latent0", [256, 1], 256, 1, 4, "p"), ("latent2", [256], 256, 1, 4, "p"),
    ("up4.up", [256, 1], 2056, 1, 4, "u"), ("up4.c0", [256, 1, 257], 256, 1, 3, "p"), ("up4.c1", [256], 256, 1, 3, "p"),
    ("up3.up", [256, 1], 2056, 1, 3, "u"), ("up3.c0", [128, 1, 257], 128, 1, 2, "p"), ("up3.c1", [128], 128, 1, 2, "p"),
    ("up2.up", [128, 1], 1032, 1, 2, "u"), ("up2.c0", [128, 1, 129], 128, 1, 1, "p"), ("up2.c1", [128], 128, 1, 1, "p"),
    ("up1.up", [128, 1], 1032, 1, 1, "u"), ("up1.c0", [64, 1, 129], 64, 1, 0, "p"), ("up1.c1", [64], 64, 1, 0, "p"),
    ("last", [64, 5], 4, 1, 0, "p"),
]


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
    ap.add_argument("--grid", type=int, nargs=3, default=[80, 320, 320])
    ap.add_argument("--only", default=None)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default=os.environ.get("SR3D_LAYER_DTYPE", "fp32"), help="storage type of the activations")
    ap.add_argument("--wgrad-only", action="store_true")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = "cuda:0"
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    print(f"{'layer':10s} {'Cin':>4s} {'Cout':>5s} s lvl | {'fwd ms':>8s} {'TF':>6s} | {'dgrad ms':>8s} {'TF':>6s} | {'wgrad ms':>8s} {'TF':>6s}")
    for name, cs, cout, stride, lvl, kind in LAYERS:
        if args.only and args.only not in name:
            continue
        Z, Y, X = [g >> lvl for g in args.grid]
        B = args.batch
        srcs = [(torch.rand(B, c, Z, Y, X, device=dev) - 0.5).to(dt) for c in cs]
        need = [c > 5 or (name != "conv0" and c > 1 and not (name == "last" and c == 5)) for c in cs]
        for s, n in zip(srcs, need):
            s.requires_grad_(n)
        cin = sum(cs)
        wf = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).requires_grad_(True)
        oz, oy, ox = [(g - 1) // stride + 1 for g in (Z, Y, X)]
        flops = 2.0 * 27 * cin * cout * oz * oy * ox * B * (2 if kind == "g" else 1)
        if kind == "g":
            wg = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).requires_grad_(True)
            bg = torch.zeros(cout, device=dev, requires_grad=True)
            fwd = lambda: ops.gated_conv3d_act(srcs, wf, wg, None, bg, act="relu", stride=stride)  # noqa: E731
        elif kind == "u":
            bias = torch.zeros(cout, device=dev, requires_grad=True)
            fwd = lambda: ops.conv3d_act(srcs, wf, bias, act="lrelu", unshuffle=True)  # noqa: E731
        else:
            fwd = lambda: ops.conv3d_act(srcs, wf, None, act="lrelu", stride=stride)  # noqa: E731
        t_f = float("nan")
        if not args.wgrad_only:
            with torch.no_grad():
                t_f = timed(fwd, args.iters)
        y = gy = None
        # dgrad only / wgrad only through the library entry points
        import ctypes as C
        from sr3d_amd import _lib as L
        desc = L.conv_desc(B, cin, cout, Z, Y, X, stride, dt)
        if kind == "g":
            dys = [torch.rand(B, cout, oz, oy, ox, device=dev).to(dt), torch.rand(B, cout, oz, oy, ox, device=dev).to(dt)]
            wgt = wg
        else:
            dys = [torch.rand(B, cout, oz, oy, ox, device=dev).to(dt)]
            wgt = None
        det = [s.detach() for s in srcs]
        t_d = float("nan")
        dflops = 0.0
        if any(need) and not args.wgrad_only:
            t_d = timed(lambda: ops._bwd_data(desc, det, need, dys, wf.detach(), None if wgt is None else wgt.detach()), args.iters)
            cneed = sum(c for c, n in zip(cs, need) if n)
            dflops = flops * cneed / cin
        t_w = timed(lambda: ops._bwd_weight(desc, det, dys), args.iters)
        if not args.wgrad_only:
            tot["fwd"][0] += t_f; tot["fwd"][1] += flops
        if any(need) and not args.wgrad_only:
            tot["dgrad"][0] += t_d; tot["dgrad"][1] += dflops
        tot["wgrad"][0] += t_w; tot["wgrad"][1] += flops
        tf = lambda fl, ms: fl / (ms * 1e-3) / 1e12 if ms == ms and ms > 0 else 0.0  # noqa: E731
        print(f"{name:10s} {cin:4d} {cout:5d} {stride} {lvl:3d} | {t_f:8.2f} {tf(flops, t_f):6.1f} | {t_d:8.2f} {tf(dflops, t_d):6.1f} | {t_w:8.2f} {tf(flops, t_w):6.1f}")
        del srcs, y, gy, dys, det
        torch.cuda.empty_cache()
    for k, (ms, fl) in tot.items():
        print(f"total {k}: {ms:.1f} ms, {fl / (ms * 1e-3) / 1e12 if ms else 0:.1f} TF")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Why did the fp32 hipGraph replay leave the eager trajectory?  ("What's weak" 1.)

Runs, in ONE process (the library reads SR3D_DEBUG_MEMSET_NODE once), at default.yml widths on a grid where the default
dispatch takes the split-f16 kernels (HR 32x64x64):

  * whole step: N eager steps against N replays of the captured step, memory dirtied with NaN patterns first;
  * single layers (forward + input / weight / bias gradients through autograd) captured into a graph of their own and
    replayed on NEW inputs against the eager result on the same inputs -- localises a divergence to a launch sequence.

    python tools/graph_diag.py            # the library as built (control words zeroed by a kernel)
    SR3D_DEBUG_MEMSET_NODE=1 python ...   # round 3's behaviour: hipMemsetAsync -> memset nodes in the graph

Prints one JSON line per check."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import sr3d_amd  # noqa: E402
from bench import make_config, synthetic_batch  # noqa: E402

DEV = torch.device("cuda:0")
MODE = "memset_node" if os.environ.get("SR3D_DEBUG_MEMSET_NODE", "0") not in ("", "0") else "zero_kernel"


def dirty(gb=6):
    """fill and free `gb` GB with a NaN bit pattern: recycled blocks (and pages the driver hands back) are dirty"""
    t = torch.full((gb * (1 << 28),), float("nan"), device=DEV)
    torch.cuda.synchronize()
    del t


def whole_step(hr, steps, loss_name):
    cfg = make_config(loss_name)
    scale = 4
    x, b, y = synthetic_batch(1, hr, scale, 1234, DEV)
    out = {}
    for kind in ("eager", "graph", "graph_again"):
        torch.manual_seed(42)
        model = sr3d_amd.make_model(cfg).to(DEV)
        loss_fn = sr3d_amd.make_loss(cfg)
        opt = sr3d_amd.FlatAdam(model.parameters(), lr=1e-4, capturable=kind != "eager")
        dirty()
        if kind == "eager":
            losses = []
            for _ in range(steps):
                loss = loss_fn(model(x, b), y, b)
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        else:
            g = sr3d_amd.GraphedTrainStep(model, loss_fn, opt, x, b, y)
            dirty()
            losses = [float(g(x, b, y)) for _ in range(steps)]
        out[kind] = {"losses": losses, "param": opt.flat_param.detach().clone()}
        del model, opt
        torch.cuda.empty_cache()
    ref = out["eager"]["param"].double()
    rec = {"check": "whole_step", "mode": MODE, "hr": hr, "loss": loss_name, "eager": out["eager"]["losses"]}
    for k in ("graph", "graph_again"):
        p = out[k]["param"].double()
        rec[k] = out[k]["losses"]
        rec[k + "_param_equal"] = bool(torch.equal(out[k]["param"], out["eager"]["param"]))
        rec[k + "_param_relerr"] = float((p - ref).norm() / ref.norm())
        rec[k + "_update_relerr"] = None
    print(json.dumps(rec), flush=True)


def layer_graph(name, fn, make_inputs, n=3):
    """fn(*inputs) -> tuple of tensors.  Capture on inputs(0), replay on inputs(1..n), compare with eager on the same"""
    static = make_inputs(0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(*static)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        outs = fn(*static)
    worst, bad = 0.0, []
    for i in range(1, n + 1):
        new = make_inputs(i)
        for s, v in zip(static, new):
            s.detach().copy_(v.detach())
        g.replay()
        torch.cuda.synchronize()
        got = [o.detach().clone() for o in outs]
        ref = fn(*new)
        torch.cuda.synchronize()
        for j, (a, r) in enumerate(zip(got, ref)):
            if not torch.equal(a, r):
                e = float((a.double() - r.double()).norm() / r.double().norm().clamp_min(1e-300))
                worst = max(worst, e)
                bad.append((i, j, e))
    print(json.dumps({"check": "layer", "mode": MODE, "layer": name, "bit_equal": not bad, "worst_relerr": worst,
                      "mismatches(replay,output,relerr)": bad[:8]}), flush=True)


def conv_case(cin, cout, grid, stride=1, gated=False, act="lrelu", unshuffle=False, bias=False):
    Z, Y, X = grid

    def make_inputs(i):
        g = torch.Generator().manual_seed(1000 + i)
        x = (torch.rand(1, cin, Z, Y, X, generator=g) - 0.3).to(DEV).requires_grad_(True)
        w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * (0.05 * (1 + i))).to(DEV).requires_grad_(True)   # max |w| moves
        w2 = (torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05).to(DEV).requires_grad_(True)
        bb = (torch.randn(cout, generator=g) * 0.1).to(DEV).requires_grad_(True)
        oz, oy, ox = [(v - 1) // stride + 1 for v in grid]
        oshape = (1, cout // 8, 2 * oz, 2 * oy, 2 * ox) if unshuffle else (1, cout, oz, oy, ox)
        gy = ((torch.rand(oshape, generator=g) - 0.5) * (10.0 ** i)).to(DEV)                                  # max |dy| moves
        return x, w, w2, bb, gy

    def fn(x, w, w2, bb, gy):
        if gated:
            yv = sr3d_amd.ops.gated_conv3d_act([x], w, w2, None, bb, act=act, stride=stride)
            ps = [x, w, w2, bb]
        else:
            yv = sr3d_amd.ops.conv3d_act([x], w, bb if bias else None, act=act, stride=stride, unshuffle=unshuffle)
            ps = [x, w] + ([bb] if bias else [])
        grads = torch.autograd.grad(yv, ps, gy)
        return (yv,) + tuple(grads)

    return fn, make_inputs


def main():
    print(json.dumps({"mode": MODE, "device": torch.cuda.get_device_name(0)}), flush=True)
    fn, mk = conv_case(64, 64, (32, 64, 64))
    layer_graph("plain 64->64 lrelu @32x64x64", fn, mk)
    fn, mk = conv_case(5, 64, (32, 64, 64), gated=True, act=None)
    layer_graph("gated 5->64 (conv0) @32x64x64", fn, mk)
    fn, mk = conv_case(65, 128, (32, 64, 64), stride=2, gated=True, act="relu")
    layer_graph("gated stride-2 65->128 @32x64x64", fn, mk)
    fn, mk = conv_case(129, 1032, (16, 32, 32), unshuffle=True, bias=True)
    layer_graph("unshuffle 129->1032 @16x32x32", fn, mk)
    whole_step((32, 64, 64), 4, "l1")
    whole_step((32, 64, 64), 3, "mixed")
    if os.environ.get("GRAPH_DIAG_FULL", "0") != "0":
        whole_step((80, 320, 320), 3, "l1")


if __name__ == "__main__":
    main()

"""torch.autograd.Functions over the libsr3d C ABI (host side of the hot path).

PyTorch owns every tensor; the library only sees raw device pointers and the
current HIP stream.  Each Function is one fused HIP forward and its hand-written
backward -- nothing here falls back to ATen convolution.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import _lib as L


# Test hook (tests/test_gpu_default_width.py): when set to a list, every fused activation appends the branch it took
# per element (bool tensor on the CPU, in forward order).  The CPU oracle can then be evaluated with the SAME decisions
# (oracle/ref_cpu.py:_act), which removes the only non-smooth step from a whole-model comparison.  Never set in
# production code: it costs a device-to-host copy per layer.
KINK_LOG: Optional[list] = None


# Weight gradient and input gradient of one layer only share their inputs: on the small grids of U-Net levels 2-4 (a few
# dozen to a few hundred workgroups per kernel, fewer than the 256 CUs) they are launched on two HIP streams so that
# together they fill the chip; on the large grids every kernel fills it alone and a second stream buys nothing.
_SIDE_STREAMS: dict = {}
# Weight gradient on a second HIP stream next to the input gradient, up to this many voxels.  OFF by default since the
# split-f16 kernels: both are then power- / LDS-bound kernels that cannot share a CU (80 + 120 KB of LDS), and running
# them side by side made the 80x320x320 step 12-15 ms SLOWER (266 vs 279-282 ms as a hipGraph replay).  With the fp32
# kernels (SR3D_SPLIT_F16=0) 1,100,000 was worth 2 ms.
CONCURRENT_WGRAD_MAX_VOXELS = int(os.environ.get("SR3D_CONCURRENT_WGRAD_MAX_VOXELS", "0"))


def _side_stream(device) -> "torch.cuda.Stream":
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device)
    return st


def _grads_two_streams(desc, srcs, needs, dys, w_feat, w_gate, want_w: bool, bias_of, x_amax=None, dy_amax=None, boxes=None):
    """(dxs, dw, [bias grads]) with the weight-side work on a side stream when the layer's grid is small"""
    vox = desc.Z * desc.Y * desc.X * desc.B
    if not want_w or not any(needs) or vox > CONCURRENT_WGRAD_MAX_VOXELS or L.PROFILING:
        dxs = _bwd_data(desc, srcs, needs, dys, w_feat, w_gate, boxes)
        dw = _bwd_weight(desc, srcs, dys, x_amax, dy_amax) if want_w else None
        return dxs, dw, [_bias_grad(t) if t is not None else None for t in bias_of]
    cur = torch.cuda.current_stream(dys[0].device)
    side = _side_stream(dys[0].device)
    side.wait_stream(cur)                      # dys / srcs were produced on the current stream
    with torch.cuda.stream(side):
        dw = _bwd_weight(desc, srcs, dys, x_amax, dy_amax)
        dbs = [_bias_grad(t) if t is not None else None for t in bias_of]
    dxs = _bwd_data(desc, srcs, needs, dys, w_feat, w_gate, boxes)
    cur.wait_stream(side)                      # join: everything returned is ready in current-stream order
    for t in [dw] + dbs:
        if t is not None:
            t.record_stream(cur)
    return dxs, dw, dbs


def _out_dim(z: int, s: int) -> int:
    return (z - 1) // s + 1


def _empty(shape, like: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    return torch.empty(shape, dtype=dtype, device=like.device)


def _check_srcs(srcs: Sequence[torch.Tensor]):
    if not 1 <= len(srcs) <= 4:
        raise ValueError("a conv input is a concat of 1..4 tensors")
    s0 = srcs[0]
    if s0.dim() != 5:
        raise ValueError("expected (B, C, z, y, x) tensors")
    for s in srcs[1:]:
        if s.shape[0] != s0.shape[0] or s.shape[2:] != s0.shape[2:]:
            raise ValueError(f"concat operands disagree: {tuple(s.shape)} vs {tuple(s0.shape)}")
        if s.dtype != s0.dtype:
            raise TypeError(f"concat operands disagree in element type: {s.dtype} vs {s0.dtype}")
    if s0.dtype not in L.DTYPE_CODE:
        raise TypeError(f"activations are float32 or bfloat16 on this engine (got {s0.dtype})")
    return s0.shape[0], sum(int(s.shape[1]) for s in srcs), tuple(s0.shape[2:])


def pack_weights(desc: L.ConvDesc, kind: int, w_feat: torch.Tensor, w_gate: Optional[torch.Tensor]) -> torch.Tensor:
    nbytes = L.lib.sr3d_packed_weight_bytes(C.byref(desc), kind)
    if nbytes == 0:
        raise RuntimeError("sr3d_packed_weight_bytes: " + L.lib.sr3d_last_error().decode())
    wp = torch.empty(nbytes // 4, dtype=torch.float32, device=w_feat.device)
    L.check(L.lib.sr3d_pack_weights(C.byref(desc), kind, L.dev_ptr(w_feat, "weight"), L.dev_ptr(w_gate, "gate weight"),
                                    L.dev_ptr(wp), L.stream_ptr()), "sr3d_pack_weights")
    return wp


def _bias_grad(dpre: torch.Tensor) -> torch.Tensor:
    B, Cc = dpre.shape[0], dpre.shape[1]
    vox = dpre[0, 0].numel()
    ws = torch.empty(max(1, L.lib.sr3d_bias_grad_workspace_bytes(B, Cc, vox) // 4), dtype=torch.float32,
                     device=dpre.device)
    db = _empty((Cc,), dpre)
    L.check(L.lib.sr3d_bias_grad(L.dev_ptr(dpre, "dpre", dpre.dtype), B, Cc, vox, L.dev_ptr(db), L.dev_ptr(ws),
                                 L.DTYPE_CODE[dpre.dtype], L.stream_ptr()), "sr3d_bias_grad")
    return db


def _amax_slots(n: int, like: torch.Tensor) -> torch.Tensor:
    """zeroed slots for the operand maxima a kernel exports as a by-product (include/sr3d.h: x_absmax / absmax_out)"""
    return torch.zeros(64 * n, dtype=torch.int32, device=like.device)


def _raw_ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _bwd_weight(desc: L.ConvDesc, srcs, dys, x_amax=None, dy_amax=None) -> torch.Tensor:
    n_total = sum(int(d.shape[1]) for d in dys)
    nbytes = L.lib.sr3d_conv3d_bwd_weight_workspace_bytes(C.byref(desc), n_total)
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=dys[0].device)
    dw = _empty((n_total, desc.Cin, 3, 3, 3), dys[0])          # weight gradients are fp32 in both storage modes
    dt = L.torch_dtype(desc)
    L.check(L.lib.sr3d_conv3d_bwd_weight(C.byref(desc), L.slices(srcs, "x_srcs", dt), len(srcs),
                                         L.slices(dys, "dy_srcs", dt), len(dys), L.dev_ptr(dw), L.dev_ptr(ws), nbytes,
                                         _raw_ptr(x_amax), _raw_ptr(dy_amax), L.stream_ptr()), "sr3d_conv3d_bwd_weight")
    return dw


# activation backward of LeakyReLU layers fused into their consumer's input-gradient epilogue where the model asks
# for it (`defer_act_bwd`) and the kernel has the epilogue; False keeps the separate lrelu_bwd pass (tests: A/B, bit-equality)
FUSE_ACT_BWD = os.environ.get("SR3D_FUSE_ACT_BWD", "1") != "0"
# the two gradients of a skip tensor (from the next block and from the skip connection) added inside the producing gated layer's
# activation backward (sr3d_gated_act_bwd_sum) instead of by autograd in an elementwise pass of its own; False: autograd adds
FUSE_SKIP_GRAD_ADD = os.environ.get("SR3D_FUSE_SKIP_GRAD_ADD", "1") != "0"


class _ActBox:
    """Shared by a LeakyReLU layer that deferred its activation backward (``defer_act_bwd=True``) and the convolution that
    consumes its output: when the consumer's input-gradient kernel has the fused epilogue (sr3d_conv3d_bwd_data_act), what
    it hands back through autograd is already dL/dpre of the producer (``done``), with max |dL/dpre| in ``amax``."""
    __slots__ = ("done", "amax", "unshuffle")

    def __init__(self, unshuffle: bool = False):
        self.done, self.amax, self.unshuffle = False, None, unshuffle


def _bwd_data(desc: L.ConvDesc, srcs, needs: Sequence[bool], dys, w_feat, w_gate, boxes=None) -> List[Optional[torch.Tensor]]:
    if not any(needs):
        return [None] * len(srcs)
    nbytes = L.lib.sr3d_conv3d_bwd_data_workspace_bytes(C.byref(desc), len(dys))
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=dys[0].device)
    outs: List[Optional[torch.Tensor]] = [torch.empty_like(s) if n else None for s, n in zip(srcs, needs)]
    dsts = [o if o is not None else (int(s.shape[1]), None) for o, s in zip(outs, srcs)]
    dt = L.torch_dtype(desc)
    dst_arr = L.slices(dsts, "dx_dsts", dt)
    # activation backward of a source layer fused into the epilogue (one slice at most: srcs[i] IS that layer's output y)
    fuse = next((i for i, (bx, n) in enumerate(zip(boxes or [], needs)) if bx is not None and n), None) if FUSE_ACT_BWD else None
    act = L.ACT_CODE["lrelu"] | (L.ACT_UNSHUFFLE if (fuse is not None and boxes[fuse].unshuffle) else 0)
    if fuse is not None and L.lib.sr3d_conv3d_bwd_data_fuses_act(C.byref(desc), len(dys), dst_arr, len(dsts), fuse, act):
        box = boxes[fuse]
        box.amax = _amax_slots(1, dys[0]) if dt == torch.float32 else None
        # (an unshuffle producer: outs[fuse] has the slice's shape for autograd's sake, but holds that layer's dL/dpre in ITS
        #  layout -- 8 C channels on the coarse grid, the same number of elements; Conv3dAct.backward views it back)
        L.check(L.lib.sr3d_conv3d_bwd_data_act(C.byref(desc), L.slices(dys, "dy_srcs", dt), len(dys), L.dev_ptr(w_feat),
                                               L.dev_ptr(w_gate), dst_arr, len(dsts), fuse, L.dev_ptr(srcs[fuse], "act_y", dt),
                                               act, _raw_ptr(box.amax), L.dev_ptr(ws), nbytes, L.stream_ptr()),
                "sr3d_conv3d_bwd_data_act")
        box.done = True
        return outs
    L.check(L.lib.sr3d_conv3d_bwd_data(C.byref(desc), L.slices(dys, "dy_srcs", dt), len(dys), L.dev_ptr(w_feat),
                                       L.dev_ptr(w_gate), dst_arr, len(dsts), L.dev_ptr(ws), nbytes,
                                       L.stream_ptr()), "sr3d_conv3d_bwd_data")
    return outs


class Conv3dAct(torch.autograd.Function):
    """y = act(conv3d(cat(srcs); W) + bias) [-> unshuffle_voxels(., 2)]

    Stands in for nn.Conv3d (+ nn.LeakyReLU) (+ VoxelUnshuffle) of the reference:
    pytorch/model/unet.py:99-108 (UpBlock.up), :72-97 (UpBlock.convs), :192-199
    (latent), :240-246 (last); custom_conv.py:111-116."""

    @staticmethod
    def forward(ctx, weight, bias, act: Optional[str], stride: int, unshuffle: bool, defer_act_bwd: bool, out_fp32: bool, *srcs):
        # sources that are outputs of LeakyReLU layers which deferred their activation backward to THIS layer's input gradient
        ctx.src_boxes = [getattr(s, "_sr3d_act_box", None) for s in srcs]
        srcs = [s.contiguous() for s in srcs]
        B, cin, (Z, Y, X) = _check_srcs(srcs)
        cout = int(weight.shape[0])
        if tuple(weight.shape[1:]) != (cin, 3, 3, 3):
            raise ValueError(f"weight {tuple(weight.shape)} does not match {cin} input channels / 3x3x3")
        weight = weight.contiguous()
        dt = srcs[0].dtype           # storage type of the activations (float32, or bfloat16: model/unet.py storage_dtype)
        desc = L.conv_desc(B, cin, cout, Z, Y, X, stride, dt)
        wp = pack_weights(desc, L.PACK_FWD_UNSHUFFLE if unshuffle else L.PACK_FWD, weight, None)
        oz, oy, ox = _out_dim(Z, stride), _out_dim(Y, stride), _out_dim(X, stride)
        # out_fp32 (bf16 storage only): the output leaves the engine as fp32 -- the network's prediction (model/unet.py)
        out_fp32 = bool(out_fp32) and dt == torch.bfloat16
        if out_fp32 and (unshuffle or stride != 1 or act is not None):
            raise ValueError("out_fp32 is the un-rounded output of a plain stride-1 layer without activation (`last`)")
        ydt = torch.float32 if out_fp32 else dt
        if unshuffle:
            y = _empty((B, cout // 8, 2 * oz, 2 * oy, 2 * ox), srcs[0], dt)
        else:
            y = _empty((B, cout, oz, oy, ox), srcs[0], ydt)
        # max |x| per slice for the weight gradient, where the forward kernel has it as a by-product (fp32, split-f16 kernel)
        x_amax = None
        if ctx.needs_input_grad[0] and L.lib.sr3d_conv3d_fwd_exports_absmax(C.byref(desc), 0):
            x_amax = _amax_slots(4, srcs[0])
        L.check(L.lib.sr3d_conv3d_fwd(C.byref(desc), L.slices(srcs, "x_srcs", dt), len(srcs), L.dev_ptr(wp),
                                      L.dev_ptr(bias, "bias"), L.dev_ptr(y, "y", ydt), L.ACT_CODE[act] | (L.ACT_OUT_F32 if out_fp32 else 0),
                                      int(bool(unshuffle)), _raw_ptr(x_amax), L.stream_ptr()), "sr3d_conv3d_fwd")
        ctx.x_amax = x_amax
        if KINK_LOG is not None and act is not None:
            KINK_LOG.append((y > 0).cpu())
        ctx.desc, ctx.act, ctx.unshuffle, ctx.has_bias, ctx.nsrc = desc, act, unshuffle, bias is not None, len(srcs)
        ctx.save_for_backward(weight, y if act is not None else None, *srcs)
        # `defer_act_bwd`: the caller guarantees that y feeds exactly ONE engine convolution; that layer's input
        # gradient then stores dL/dy * lrelu'(y) directly and this layer's lrelu_bwd pass never runs (model/unet.py)
        ctx.act_box = None
        if defer_act_bwd and act == "lrelu" and any(ctx.needs_input_grad):
            ctx.act_box = y._sr3d_act_box = _ActBox(unshuffle=bool(unshuffle))
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, y, *srcs = ctx.saved_tensors
        desc = ctx.desc
        dt = L.torch_dtype(desc)
        dy = dy.to(dt).contiguous()
        # max |dpre| for the weight gradient comes out of the activation-backward kernel (fp32 only)
        dy_amax = _amax_slots(1, dy) if (ctx.needs_input_grad[0] and dt == torch.float32 and ctx.act is not None) else None
        box = ctx.act_box
        if box is not None and box.done:      # the consumer's input-gradient epilogue already applied lrelu'(y)
            dpre, dy_amax = dy, (box.amax if dy_amax is not None else None)
            if ctx.unshuffle:                 # ... and wrote it in THIS layer's layout: 8 C channels on the coarse grid
                B, c, z2, y2, x2 = dy.shape
                dpre = dy.view(B, 8 * c, z2 // 2, y2 // 2, x2 // 2)
            box.done, box.amax = False, None
        elif ctx.unshuffle:
            B, c, z2, y2, x2 = dy.shape
            dpre = _empty((B, 8 * c, z2 // 2, y2 // 2, x2 // 2), dy, dt)
            if ctx.act != "lrelu":
                raise NotImplementedError("unshuffle epilogue is defined with LeakyReLU (unet.py:99-108)")
            L.check(L.lib.sr3d_unshuffle_lrelu_bwd(L.dev_ptr(dy, "dy", dt), L.dev_ptr(y, "y", dt),
                                                   L.dev_ptr(dpre, "dpre", dt), B, c, z2 // 2, y2 // 2, x2 // 2,
                                                   desc.dtype, _raw_ptr(dy_amax), L.stream_ptr()), "sr3d_unshuffle_lrelu_bwd")
        elif ctx.act == "lrelu":
            dpre = torch.empty_like(dy)
            L.check(L.lib.sr3d_lrelu_bwd(L.dev_ptr(dy, "dy", dt), L.dev_ptr(y, "y", dt), L.dev_ptr(dpre, "dpre", dt),
                                         dy.numel(), desc.dtype, _raw_ptr(dy_amax), L.stream_ptr()), "sr3d_lrelu_bwd")
        elif ctx.act is None:
            dpre = dy
        else:
            raise NotImplementedError(f"backward of plain conv with act={ctx.act}")
        needs = ctx.needs_input_grad[7:7 + ctx.nsrc]
        want_b = ctx.has_bias and ctx.needs_input_grad[1]
        dxs, dw, (db,) = _grads_two_streams(desc, srcs, needs, [dpre], weight, None, ctx.needs_input_grad[0],
                                            [dpre if want_b else None], ctx.x_amax, dy_amax, ctx.src_boxes)
        return (dw, db, None, None, None, None, None, *dxs)


class GatedConv3dAct(torch.autograd.Function):
    """y = sigmoid(conv(x; Wg) + bg) * act(conv(x; Wf) [+ bf])

    One fused kernel for ``GatedConv3d(WithSeparatedBias)`` + ``MyConvWithAct2``
    (reference custom_conv.py:119-123, 237-306)."""

    @staticmethod
    def forward(ctx, w_feat, w_gate, b_feat, b_gate, act: Optional[str], stride: int, dual: bool, *srcs):
        """``dual``: return the output TWICE (the same storage): for an output with two consumers -- the next block and the
        U-Net's skip connection -- backward then receives the two gradients separately and adds them inside
        ``sr3d_gated_act_bwd_sum`` instead of autograd adding them in an elementwise pass of its own"""
        srcs = [s.contiguous() for s in srcs]
        B, cin, (Z, Y, X) = _check_srcs(srcs)
        cout = int(w_feat.shape[0])
        if tuple(w_feat.shape) != (cout, cin, 3, 3, 3) or w_gate.shape != w_feat.shape:
            raise ValueError("gated conv: weight shapes do not match the input")
        w_feat, w_gate = w_feat.contiguous(), w_gate.contiguous()
        dt = srcs[0].dtype
        desc = L.conv_desc(B, cin, cout, Z, Y, X, stride, dt)
        wp = pack_weights(desc, L.PACK_FWD_GATED, w_feat, w_gate)
        oshape = (B, cout, _out_dim(Z, stride), _out_dim(Y, stride), _out_dim(X, stride))
        y = _empty(oshape, srcs[0], dt)
        need_bwd = any(ctx.needs_input_grad)
        # saved for backward: sigmoid(gate) and the output y itself (alive anyway as the next layer's input).  act(feat) is NOT
        # stored: sigmoid > 0, so y has its sign (the same act') and act(feat) * sigmoid = y (sr3d_gated_act_bwd, SR3D_ACT_FROM_Y):
        # one output-sized store in forward and one load in backward less per gated layer (6 GB per step at 80x320x320)
        ss = _empty(oshape, y, dt) if need_bwd else None
        x_amax = None
        if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and L.lib.sr3d_conv3d_fwd_exports_absmax(C.byref(desc), 1):
            x_amax = _amax_slots(4, srcs[0])
        L.check(L.lib.sr3d_gated_conv3d_fwd(C.byref(desc), L.slices(srcs, "x_srcs", dt), len(srcs), L.dev_ptr(wp),
                                            L.dev_ptr(b_feat, "feature bias"), L.dev_ptr(b_gate, "gate bias"),
                                            L.dev_ptr(y, "y", dt), None,
                                            L.dev_ptr(ss, "save_s", dt), L.ACT_CODE[act], _raw_ptr(x_amax),
                                            L.stream_ptr()), "sr3d_gated_conv3d_fwd")
        ctx.x_amax = x_amax
        if KINK_LOG is not None and act is not None:
            KINK_LOG.append((y > 0).cpu())       # (the sign of act(feat): sigmoid > 0)
        ctx.desc, ctx.act, ctx.nsrc, ctx.has_bf = desc, act, len(srcs), b_feat is not None
        ctx.save_for_backward(w_feat, w_gate, y if need_bwd else None, ss, *srcs)
        if not dual:
            return y
        ctx.set_materialize_grads(False)   # (an unused alias brings None, not a tensor of zeros)
        return y, y.detach()

    @staticmethod
    def backward(ctx, dy, dy2=None):
        w_feat, w_gate, y, ss, *srcs = ctx.saved_tensors
        desc = ctx.desc
        dt = L.torch_dtype(desc)
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:                      # (neither output reached the loss)
            return (None,) * (7 + ctx.nsrc)
        dy = dy.to(dt).contiguous()
        dy2 = dy2.to(dt).contiguous() if dy2 is not None else None
        d_feat, d_gate = torch.empty_like(dy), torch.empty_like(dy)
        want_w = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        dy_amax = _amax_slots(2, dy) if (want_w and dt == torch.float32) else None
        L.check(L.lib.sr3d_gated_act_bwd_sum(L.dev_ptr(dy, "dy", dt), L.dev_ptr(dy2, "dy2", dt),
                                             L.dev_ptr(y, "y", dt), L.dev_ptr(ss, "save_s", dt),
                                             L.dev_ptr(d_feat, "d_feat", dt), L.dev_ptr(d_gate, "d_gate", dt), dy.numel(),
                                             L.ACT_CODE[ctx.act] | L.ACT_FROM_Y, desc.dtype, _raw_ptr(dy_amax), L.stream_ptr()),
                "sr3d_gated_act_bwd_sum")
        needs = ctx.needs_input_grad[7:7 + ctx.nsrc]
        dxs, dw, (dbf, dbg) = _grads_two_streams(
            desc, srcs, needs, [d_feat, d_gate], w_feat, w_gate, want_w,
            [d_feat if (ctx.has_bf and ctx.needs_input_grad[2]) else None, d_gate if ctx.needs_input_grad[3] else None],
            ctx.x_amax, dy_amax)
        dwf, dwg = (dw[:desc.Cout], dw[desc.Cout:]) if dw is not None else (None, None)
        return (dwf, dwg, dbf, dbg, None, None, None, *dxs)


def conv3d_act(srcs, weight, bias=None, act=None, stride=1, unshuffle=False, defer_act_bwd=False, out_fp32=False):
    return Conv3dAct.apply(weight, bias, act, stride, unshuffle, defer_act_bwd, out_fp32, *srcs)


def gated_conv3d_act(srcs, w_feat, w_gate, b_feat, b_gate, act=None, stride=1, dual=False):
    """``dual=True`` returns (y, y again): see GatedConv3dAct.forward"""
    return GatedConv3dAct.apply(w_feat, w_gate, b_feat, b_gate, act, stride, dual, *srcs)


# ---------------------------------------------------------------- PartialConv3d pieces
def pconv_mask_update(mask: torch.Tensor, stride: int, slide_winsize: float):
    """(update_mask, mask_ratio), each (Bm, 1, OZ, OY, OX) (reference custom_conv.py:203-216)"""
    mask = mask.detach().contiguous()
    Bm, Cm, Z, Y, X = mask.shape
    shape = (Bm, 1, _out_dim(Z, stride), _out_dim(Y, stride), _out_dim(X, stride))
    upd, ratio = _empty(shape, mask), _empty(shape, mask)
    L.check(L.lib.sr3d_pconv_mask_update(L.dev_ptr(mask), Bm, Cm, Z, Y, X, stride, float(slide_winsize), L.dev_ptr(upd),
                                         L.dev_ptr(ratio), L.stream_ptr()), "sr3d_pconv_mask_update")
    return upd, ratio


class MulMask(torch.autograd.Function):
    """x * mask with the mask broadcast over batch / channels; the mask carries no gradient"""

    @staticmethod
    def forward(ctx, x, mask):
        x, mask = x.contiguous(), mask.detach().contiguous()
        ctx.save_for_backward(mask)
        return MulMask._run(x, mask)

    @staticmethod
    def _run(x, mask):
        B, c = x.shape[0], x.shape[1]
        out = torch.empty_like(x)
        L.check(L.lib.sr3d_mul_mask(L.dev_ptr(x), L.dev_ptr(mask), L.dev_ptr(out), B, c, x[0, 0].numel(),
                                    int(mask.shape[0]), int(mask.shape[1]), L.stream_ptr()), "sr3d_mul_mask")
        return out

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return MulMask._run(g.contiguous(), mask), None


class PconvScale(torch.autograd.Function):
    """the renormalisation after the convolution (custom_conv.py:224-229)"""

    @staticmethod
    def forward(ctx, raw, bias, upd, ratio):
        raw = raw.contiguous()
        B, c = raw.shape[0], raw.shape[1]
        out = torch.empty_like(raw)
        L.check(L.lib.sr3d_pconv_scale(L.dev_ptr(raw), L.dev_ptr(bias, "bias"), L.dev_ptr(upd), L.dev_ptr(ratio),
                                       L.dev_ptr(out), None, B, c, raw[0, 0].numel(), int(upd.shape[0]), 0,
                                       L.stream_ptr()), "sr3d_pconv_scale")
        ctx.save_for_backward(bias, upd, ratio)
        return out

    @staticmethod
    def backward(ctx, g):
        bias, upd, ratio = ctx.saved_tensors
        g = g.contiguous()
        B, c = g.shape[0], g.shape[1]
        d_raw = torch.empty_like(g)
        terms = torch.empty_like(g) if (bias is not None and ctx.needs_input_grad[1]) else None
        L.check(L.lib.sr3d_pconv_scale(L.dev_ptr(g), L.dev_ptr(bias, "bias"), L.dev_ptr(upd), L.dev_ptr(ratio),
                                       L.dev_ptr(d_raw), L.dev_ptr(terms), B, c, g[0, 0].numel(), int(upd.shape[0]), 1,
                                       L.stream_ptr()), "sr3d_pconv_scale")
        return d_raw, (_bias_grad(terms) if terms is not None else None), None, None


# ---------------------------------------------------------------- no-grad data movement
def upsample_cat(x: torch.Tensor, b: torch.Tensor, scale: int) -> torch.Tensor:
    """cat[nearest_upsample(x, scale), b]  (reference unet.py:143,254-255); inputs carry no gradient."""
    x, b = x.detach().contiguous(), b.detach().contiguous()
    B, c = x.shape[0], x.shape[1]
    Z, Y, X = b.shape[2:]
    if tuple(x.shape[2:]) != (Z // scale, Y // scale, X // scale) or b.shape[1] != 1 or b.shape[0] != B:
        raise ValueError(f"upsample_cat: x {tuple(x.shape)} x{scale} does not match mask {tuple(b.shape)}")
    out = _empty((B, c + 1, Z, Y, X), x)
    L.check(L.lib.sr3d_upsample_cat(L.dev_ptr(x), L.dev_ptr(b), L.dev_ptr(out), B, c, Z, Y, X, scale, L.stream_ptr()),
            "sr3d_upsample_cat")
    return out


def avgpool2(b: torch.Tensor) -> torch.Tensor:
    """nn.AvgPool3d(2, 2) of the (B,1,Z,Y,X) building mask (reference unet.py:156)."""
    b = b.detach().contiguous()
    B, c, Z, Y, X = b.shape
    if c != 1:
        raise ValueError("avgpool2 is the mask pyramid op: one channel")
    out = _empty((B, 1, Z // 2, Y // 2, X // 2), b)
    L.check(L.lib.sr3d_avgpool2(L.dev_ptr(b), L.dev_ptr(out), B, Z, Y, X, L.stream_ptr()), "sr3d_avgpool2")
    return out


def near_wall_mask(b: torch.Tensor) -> torch.Tensor:
    """calc_mask_near_build_wall (reference loss_maker.py:57-83)."""
    b = b.detach().contiguous()
    B, c, Z, Y, X = b.shape
    if c != 1:
        raise ValueError("near_wall_mask expects a (B,1,Z,Y,X) mask")
    out = torch.empty_like(b)
    L.check(L.lib.sr3d_near_wall(L.dev_ptr(b), L.dev_ptr(out), B, Z, Y, X, L.stream_ptr()), "sr3d_near_wall")
    return out


def preprocess(x: torch.Tensor, means: Sequence[float], stds: Sequence[float], scaling: Optional[float] = None,
               clip: bool = True, nan_value: float = 0.0, discard_z: int = 0) -> torch.Tensor:
    """the reference Dataset's normalise / clamp / NaN-fill of one (B, C, Z, Y, X) tensor in physical units, on the
    device (dataset.py:139-161, 174, 191-195); bit-identical to the CPU path of ``src/dataset.py``"""
    x = x.contiguous()
    if x.dim() != 5:
        raise ValueError("preprocess expects (B, C, Z, Y, X)")
    B, c, Z, Y, X = x.shape
    if len(means) != c or len(stds) != c:
        raise ValueError("preprocess: one mean / std per channel")
    out = torch.empty_like(x)
    m = (C.c_float * c)(*[float(v) for v in means])
    sd = (C.c_float * c)(*[float(v) for v in stds])
    L.check(L.lib.sr3d_preprocess(L.dev_ptr(x), L.dev_ptr(out), B, c, Z, Y, X, m, sd,
                                  1.0 if scaling is None else float(scaling), int(bool(clip)), float(nan_value),
                                  int(discard_z or 0), L.stream_ptr()), "sr3d_preprocess")
    return out


# ---------------------------------------------------------------- losses
class L1LossFn(torch.autograd.Function):
    """mean |p - t| and its gradient in one pass (reference loss_maker.py:194-202)."""

    @staticmethod
    def forward(ctx, p, t):
        p, t = p.contiguous(), t.detach().contiguous()
        if p.shape != t.shape:
            raise ValueError("L1: shapes differ")
        need = ctx.needs_input_grad[0]
        ws = torch.empty(L.lib.sr3d_loss_workspace_bytes(1, 1, 1, 1) // 4, dtype=torch.float32, device=p.device)
        out = _empty((1,), p)
        g = torch.empty_like(p) if need else None
        L.check(L.lib.sr3d_l1_fwd_bwd(L.dev_ptr(p), L.dev_ptr(t), p.numel(), L.dev_ptr(out), L.dev_ptr(g),
                                      L.dev_ptr(ws), L.stream_ptr()), "sr3d_l1_fwd_bwd")
        ctx.save_for_backward(g)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        return g * gout, None


class MixedLossFn(torch.autograd.Function):
    """MixedDivergenceGradientL2Loss (reference loss_maker.py:387-450).
    Returns a 4-vector (mse, grd_mse, div_mse, total).  Every component is differentiable (GradNorm takes
    per-term gradients, gradnorm.py:95-100): backward runs ONE adjoint-stencil kernel for whatever linear
    combination of the terms autograd asks for; the combination weights stay on the device."""

    @staticmethod
    def forward(ctx, p, t, b, scales, delta, w_g, w_d):
        p, t, b = p.contiguous(), t.detach().contiguous(), b.detach().contiguous()
        B, c, Z, Y, X = p.shape
        if c != 4 or t.shape != p.shape or tuple(b.shape) != (B, 1, Z, Y, X):
            raise ValueError("mixed loss expects p,t: (B,4,Z,Y,X) and masks: (B,1,Z,Y,X)")
        ws = torch.empty(L.lib.sr3d_loss_workspace_bytes(B, Z, Y, X) // 4, dtype=torch.float32, device=p.device)
        terms = _empty((4,), p)
        sc = (C.c_float * 3)(*[float(s) for s in scales])
        L.check(L.lib.sr3d_mixed_div_grad_l2_fwd_bwd(L.dev_ptr(p), L.dev_ptr(t), L.dev_ptr(b), B, Z, Y, X, sc,
                                                     float(delta), float(w_g), float(w_d), L.dev_ptr(terms),
                                                     None, L.dev_ptr(ws), L.stream_ptr()),
                "sr3d_mixed_div_grad_l2_fwd_bwd")
        ctx.args = (B, Z, Y, X, [float(s) for s in scales], float(delta), float(w_g), float(w_d))
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(p, t, ws)
        return terms

    @staticmethod
    def backward(ctx, gterms):
        p, t, ws = ctx.saved_tensors
        B, Z, Y, X, scales, delta, w_g, w_d = ctx.args
        g = gterms.to(torch.float32)
        wts = torch.stack([g[0] + g[3], g[1] + g[3] * w_g, g[2] + g[3] * w_d]).contiguous()
        dp = torch.empty_like(p)
        sc = (C.c_float * 3)(*scales)
        L.check(L.lib.sr3d_mixed_div_grad_l2_bwd(L.dev_ptr(p), L.dev_ptr(t), B, Z, Y, X, sc, delta, w_g, w_d,
                                                 L.dev_ptr(wts), L.dev_ptr(dp), L.dev_ptr(ws), L.stream_ptr()),
                "sr3d_mixed_div_grad_l2_bwd")
        return dp, None, None, None, None, None, None


# ---------------------------------------------------------------- evaluation metrics
_eval_cache: dict = {"refs": None, "meta": None, "stds": None, "out": None}


def invalidate_eval_cache() -> None:
    """forget the cached metrics launch (a hipGraph replay or an in-place kernel refilled a buffer behind autograd's back)"""
    _eval_cache.update(refs=None, meta=None, stds=None, out=None)


def _eval_hit(p, t, b, meta) -> bool:
    """the cached launch was made for exactly these three tensor OBJECTS (weak references: an address can be reused by
    the allocator for the next batch, an object cannot), unmodified since (version counters), same parameters"""
    refs = _eval_cache["refs"]
    if refs is None or _eval_cache["meta"] != meta:
        return False
    return all(r() is x for r, x in zip(refs, (p, t, b)))


def eval_metrics(p: torch.Tensor, t: torch.Tensor, b: torch.Tensor, stds: Sequence[Optional[float]],
                 delta: float = 5.0, lev: int = 0) -> torch.Tensor:
    """All evaluation metrics of the reference's test pass from ONE fused kernel (include/sr3d.h: sr3d_eval_metrics);
    returns the SR3D_EVAL_COUNT-vector on the device (index with ``_lib.EVAL_INDEX``).

    ``stds`` = (std_T, std_u, std_v, std_w); an entry may be None when the caller does not depend on it.  The result
    of the last launch is kept: the reference calls its ten metric modules one after the other on the same
    (prediction, target, mask), and every call after the first is a cache hit as long as the scales it needs agree
    with the ones the launch used (``optim_helper.evaluate`` makes the first call with the union of all scales)."""
    import weakref
    p0, t0, b0 = p, t, b
    p, t, b = p.detach().contiguous(), t.detach().contiguous(), b.detach().contiguous()
    B, c, Z, Y, X = p.shape
    if c != 4 or t.shape != p.shape or tuple(b.shape) != (B, 1, Z, Y, X):
        raise ValueError("eval_metrics expects p, t: (B,4,Z,Y,X) and masks: (B,1,Z,Y,X)")
    # (the addresses are part of the key: the engine's own kernels and hipGraph replays write through raw pointers and
    # never bump `_version`; `invalidate_eval_cache()` is called by every graph replay for the same reason)
    meta = (p0._version, t0._version, b0._version, p.data_ptr(), t.data_ptr(), b.data_ptr(), float(delta), int(lev),
            torch.cuda.current_stream().cuda_stream)
    want = [None if v is None else float(v) for v in stds]
    if _eval_hit(p0, t0, b0, meta) and all(w is None or w == h for w, h in zip(want, _eval_cache["stds"])):
        return _eval_cache["out"]
    used = [1.0 if w is None else w for w in want]
    ws = torch.empty(L.lib.sr3d_eval_metrics_workspace_bytes(B, Z, Y, X) // 4, dtype=torch.float32, device=p.device)
    out = _empty((L.EVAL_COUNT,), p)
    sc = (C.c_float * 4)(*used)
    L.check(L.lib.sr3d_eval_metrics(L.dev_ptr(p), L.dev_ptr(t), L.dev_ptr(b), B, Z, Y, X, sc, float(delta), int(lev),
                                    L.dev_ptr(out), L.dev_ptr(ws), L.stream_ptr()), "sr3d_eval_metrics")
    # entries a caller left open are NOT remembered as 1.0: a later caller that needs them must relaunch
    _eval_cache.update(refs=tuple(weakref.ref(x) for x in (p0, t0, b0)), meta=meta,
                       stds=[w if w is not None else float("nan") for w in want], out=out)
    return out


def ssim3d(img1: torch.Tensor, img2: torch.Tensor, mask: torch.Tensor, window: Sequence[float], max_val: float = 1.0,
           eps: float = 1e-7, size_average: bool = True) -> torch.Tensor:
    """masked SSIM3D of the reference (src/ssim.py:52-115) as three separable HIP passes; forward only.
    ``window``: the 1-D taps whose triple outer product is the reference's window."""
    if torch.is_grad_enabled() and (img1.requires_grad or img2.requires_grad):
        raise NotImplementedError("ssim3d is an evaluation metric on this engine: call it under torch.no_grad()")
    img1, img2, mask = img1.detach().contiguous(), img2.detach().contiguous(), mask.detach().contiguous()
    B, c, Z, Y, X = img1.shape
    if img2.shape != img1.shape or mask.shape[0] != B or tuple(mask.shape[2:]) != (Z, Y, X) or mask.shape[1] not in (1, c):
        raise ValueError("ssim3d: img1, img2 (B,C,Z,Y,X) and a mask with 1 or C channels on the same grid")
    n = len(window)
    ws = torch.empty(L.lib.sr3d_ssim3d_workspace_bytes(B, c, Z, Y, X) // 4, dtype=torch.float32, device=img1.device)
    mean = _empty((1,), img1)
    smap = None if size_average else torch.empty_like(img1)
    taps = (C.c_float * n)(*[float(v) for v in window])
    L.check(L.lib.sr3d_ssim3d(L.dev_ptr(img1), L.dev_ptr(img2), L.dev_ptr(mask), B, c, int(mask.shape[1]), Z, Y, X, taps,
                              n, float(max_val), float(eps), L.dev_ptr(mean), L.dev_ptr(smap), L.dev_ptr(ws),
                              L.stream_ptr()), "sr3d_ssim3d")
    return mean[0] if size_average else smap


def eval_shapes_ok(p: torch.Tensor, t: torch.Tensor, b: torch.Tensor) -> bool:
    """the fused evaluation pass takes (B,4,Z,Y,X) fields and a (B,1,Z,Y,X) mask"""
    return (p.dim() == 5 and p.shape[1] == 4 and t.shape == p.shape and b is not None and b.dim() == 5
            and tuple(b.shape) == (p.shape[0], 1) + tuple(p.shape[2:]))


class WeightedLpFn(torch.autograd.Function):
    """WeightedL1Loss / WeightedL2Loss (reference loss_maker.py:216-255):
    (w * sum_in(e) / (N_in + 1) + sum_out(e) / (N_out + 1)) / (w + 1) with e = |p - t| or (p - t)^2, `in` = mask 1
    (outside buildings), the mask broadcast over the channels.  The five sums come out of the fused evaluation pass;
    the gradient is one streaming kernel."""

    @staticmethod
    def forward(ctx, p, t, b, weight: float, power: int):
        p, t, b = p.contiguous(), t.detach().contiguous(), b.detach().contiguous()
        B, c = p.shape[0], p.shape[1]
        sums = eval_metrics(p, t, b, (None,) * 4)
        k = L.EVAL_SUMS
        s_all, s_in = (sums[k["abs"]], sums[k["mask_abs"]]) if power == 1 else (sums[k["sq"]], sums[k["mask_sq"]])
        n_in = c * sums[k["mask"]]
        n_out = float(p.numel()) - n_in
        w = float(weight)
        loss = (w * s_in / (n_in + 1.0) + (s_all - s_in) / (n_out + 1.0)) / (w + 1.0)
        ctx.save_for_backward(p, t, b, torch.stack([w / ((n_in + 1.0) * (w + 1.0)), 1.0 / ((n_out + 1.0) * (w + 1.0))]))
        ctx.power = power
        return loss

    @staticmethod
    def backward(ctx, gout):
        p, t, b, coef = ctx.saved_tensors
        coef = (coef * gout).to(torch.float32).contiguous()
        g = torch.empty_like(p)
        B, c = p.shape[0], p.shape[1]
        L.check(L.lib.sr3d_weighted_lp_bwd(L.dev_ptr(p), L.dev_ptr(t), L.dev_ptr(b), B, c, p[0, 0].numel(), ctx.power,
                                           L.dev_ptr(coef), L.dev_ptr(g), L.stream_ptr()), "sr3d_weighted_lp_bwd")
        return g, None, None, None, None


# ---------------------------------------------------------------- optimizer
def adam_step_(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, lr: float,
               beta1: float, beta2: float, eps: float, step: int, grad_scale: float = 1.0) -> None:
    """in-place fused Adam on flat fp32 buffers (torch.optim.Adam defaults; reference train_model.py:183)."""
    n = param.numel()
    if not (grad.numel() == exp_avg.numel() == exp_avg_sq.numel() == n):
        raise ValueError("adam_step_: buffer sizes differ")
    L.check(L.lib.sr3d_adam_step(L.dev_ptr(param), L.dev_ptr(grad), L.dev_ptr(exp_avg), L.dev_ptr(exp_avg_sq), n,
                                 float(lr), float(beta1), float(beta2), float(eps), int(step), float(grad_scale),
                                 L.stream_ptr()), "sr3d_adam_step")


def adam_step_device_counter_(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                              lr: float, beta1: float, beta2: float, eps: float, step_counter: torch.Tensor,
                              scalars: torch.Tensor, grad_scale: float = 1.0) -> None:
    """the same update with the step number in device memory (int32 tensor, incremented by the call): capturable"""
    n = param.numel()
    if not (grad.numel() == exp_avg.numel() == exp_avg_sq.numel() == n):
        raise ValueError("adam_step_device_counter_: buffer sizes differ")
    if step_counter.dtype != torch.int32 or not step_counter.is_cuda or scalars.numel() < 2:
        raise ValueError("adam_step_device_counter_: step_counter must be an int32 GPU tensor, scalars 2 floats")
    L.check(L.lib.sr3d_adam_step_device_counter(L.dev_ptr(param), L.dev_ptr(grad), L.dev_ptr(exp_avg),
                                                L.dev_ptr(exp_avg_sq), n, float(lr), float(beta1), float(beta2),
                                                float(eps), C.c_void_p(step_counter.data_ptr()), L.dev_ptr(scalars),
                                                float(grad_scale), L.stream_ptr()), "sr3d_adam_step_device_counter")

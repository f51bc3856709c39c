"""The split-f16 stride-1 convolution kernel (csrc/sr3d_hconv.hip): fp32 operands as two fp16 halves, three f16 MFMAs
per product group, fp32 accumulation -- against fp64 references (pytest -m gpu, on the MI355X box).

SR3D_SPLIT_F16=2 forces the kernel for every eligible layer (by default only launches that fill the chip take it);
the switch is read per call, so the fixtures flip it in-process.  Tolerance: 1e-5 normwise as everywhere else; the
measured error is ~4e-7, that of the fp32 MFMA kernels."""
import pytest
import torch
import torch.nn.functional as F

from helpers import relerr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import sr3d_amd
    return sr3d_amd


@pytest.fixture()
def forced(monkeypatch):
    monkeypatch.setenv("SR3D_SPLIT_F16", "2")


def _ref_and_inputs(cs, cout, grid, kind, scale, seed, batch=2, stride=1):
    g = torch.Generator().manual_seed(seed)
    Z, Y, X = grid
    xs = [((torch.rand(batch, c, Z, Y, X, generator=g) - 0.3) * scale).double().requires_grad_(c > 1) for c in cs]
    cin = sum(cs)
    wf = (torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05).double().requires_grad_(True)
    wg = (torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.05).double().requires_grad_(True)
    bias = (torch.randn(cout, generator=g) * 0.1 * scale).double().requires_grad_(True)
    x = torch.cat(xs, 1)
    if kind == "gated":
        pre_f, pre_g = F.conv3d(x, wf, None, stride=stride, padding=1), F.conv3d(x, wg, bias, stride=stride, padding=1)
        ref = torch.sigmoid(pre_g) * torch.relu(pre_f)
    else:
        pre_f = F.conv3d(x, wf, bias, stride=stride, padding=1)
        ref = F.leaky_relu(pre_f, 0.01)
    gy = (torch.rand(ref.shape, generator=g) - 0.5).double()
    # keep the comparison on the smooth part: no gradient through pre-activations within 1e-4 of the kink
    gy = gy * (pre_f.detach().abs() > 1e-4 * scale)
    ref.backward(gy)
    return xs, wf, wg, bias, ref, gy


# (source channels, Cout, grid, kind, scale): ragged grids, K not a multiple of 16, a concat boundary inside a chunk,
# <= 32-row and 33-row row blocks, 1 .. 4 remainder rows, tiny and large magnitudes (the per-call power-of-two scaling)
CASES = [
    ([64], 64, (8, 16, 64), "plain", 1.0),
    ([64, 1, 65], 48, (6, 10, 40), "plain", 1.0),
    ([33], 72, (5, 7, 33), "plain", 1.0),
    ([40], 130, (3, 9, 70), "plain", 300.0),
    ([64], 64, (4, 8, 32), "plain", 1e-7),
    ([128], 32, (4, 8, 32), "gated", 1.0),
    ([36, 1], 40, (5, 6, 35), "gated", 1.0),
    ([4, 1], 64, (6, 10, 40), "gated", 1.0),     # conv0: K = 5 is ONE half chunk (4 taps x 8 channels per MFMA, 7 phases)
    ([64, 5], 4, (6, 10, 40), "plain", 1.0),     # `last`: the input gradient has K = 4
    ([128, 1], 40, (4, 8, 32), "plain", 1.0),    # 8 full chunks + a tail of one channel (round 4: im2col form, K = its 27 taps)
    ([32, 2], 130, (4, 8, 36), "plain", 1.0),    # ... of two channels, forward; 130 rows: the input gradient's K = 130 has one too
]


@pytest.mark.parametrize("cs,cout,grid,kind,scale", CASES)
def test_split_f16_layers_vs_fp64(eng, forced, cs, cout, grid, kind, scale):
    xs, wf, wg, bias, ref, gy = _ref_and_inputs(cs, cout, grid, kind, scale, seed=sum(cs) + cout)
    dev = lambda t: t.detach().float().to(DEV)   # noqa: E731
    xd = [dev(x).requires_grad_(x.requires_grad) for x in xs]
    wfd, wgd, bd = dev(wf).requires_grad_(True), dev(wg).requires_grad_(True), dev(bias).requires_grad_(True)
    if kind == "gated":
        y = eng.ops.gated_conv3d_act(xd, wfd, wgd, None, bd, act="relu", stride=1)
    else:
        y = eng.ops.conv3d_act(xd, wfd, bd, act="lrelu", stride=1)
    assert relerr(y, ref) < TOL
    y.backward(dev(gy))
    for a, b in zip(xd, xs):
        if b.requires_grad:
            assert relerr(a.grad, b.grad) < TOL
    assert relerr(wfd.grad, wf.grad) < TOL and relerr(bd.grad, bias.grad) < TOL
    if kind == "gated":
        assert relerr(wgd.grad, wg.grad) < TOL


# stride 2 (csrc/sr3d_hconv_s2.hip: parity classes): odd and even grids, a concat boundary inside a chunk, row blocks of
# 32 / 40 / 64 / 130 rows, the mask-like single channel that needs no gradient
S2_CASES = [
    ([64, 1], 64, (8, 16, 64), "gated", 1.0),
    ([33], 40, (5, 7, 33), "plain", 1.0),
    ([48], 130, (6, 9, 70), "plain", 50.0),
    ([40, 1, 24], 32, (7, 8, 34), "gated", 1e-6),
    ([40], 48, (6, 10, 72), "plain", 1.0),        # coarse rows of 36: quad dY loads with a partial last x tile
    # round 4, the x-paired kernels (forward: X % 4 == 0; input gradient: coarse X % 4 == 0):
    ([20, 1], 36, (7, 9, 40), "gated", 1.0),      # odd z / y under the paired forward; 72 rows = a 64-row block + an 8-row one
    ([24], 40, (5, 9, 71), "plain", 1.0),         # odd fine X over coarse rows of 36: paired gradient, its odd class one shorter, scalar stores
    ([17, 1], 20, (4, 6, 36), "plain", 3.0),      # paired forward, coarse X = 18: class-form gradient next to it
]


@pytest.mark.parametrize("cs,cout,grid,kind,scale", S2_CASES)
def test_split_f16_stride2_layers_vs_fp64(eng, forced, cs, cout, grid, kind, scale):
    xs, wf, wg, bias, ref, gy = _ref_and_inputs(cs, cout, grid, kind, scale, seed=7 * sum(cs) + cout, stride=2)
    dev = lambda t: t.detach().float().to(DEV)   # noqa: E731
    xd = [dev(x).requires_grad_(x.requires_grad) for x in xs]
    wfd, wgd, bd = dev(wf).requires_grad_(True), dev(wg).requires_grad_(True), dev(bias).requires_grad_(True)
    if kind == "gated":
        y = eng.ops.gated_conv3d_act(xd, wfd, wgd, None, bd, act="relu", stride=2)
    else:
        y = eng.ops.conv3d_act(xd, wfd, bd, act="lrelu", stride=2)
    assert y.shape == ref.shape
    assert relerr(y, ref) < TOL
    y.backward(dev(gy))
    for a, b in zip(xd, xs):
        if b.requires_grad:
            assert relerr(a.grad, b.grad) < TOL
    assert relerr(wfd.grad, wf.grad) < TOL and relerr(bd.grad, bias.grad) < TOL
    if kind == "gated":
        assert relerr(wgd.grad, wg.grad) < TOL


# the split-f16 weight gradient (csrc/sr3d_hwgrad.hip) takes stride-1 layers with X % 8 == 0: 32-row and 64-row
# workgroups, a last row block with 2 rows, 1-4 input channels beyond a multiple of 32 (few-channel kernel next to it),
# a partial last channel block, two dY slices (gated), more than one x segment, several splits, batch 2
@pytest.mark.parametrize("cs,cout,grid,kind", [
    ([64], 24, (4, 12, 32), "plain"),
    ([32, 1, 33], 130, (5, 9, 72), "plain"),
    ([40], 64, (7, 30, 48), "plain"),
    ([64, 2], 36, (3, 26, 40), "gated"),
    ([64, 5], 4, (4, 12, 32), "plain"),        # the `last` layer's shape: 4 rows (round 4: hwgrad_fc with x and dY exchanged, 80-row workgroups)
    ([96, 3], 3, (4, 12, 32), "plain"),        # ... 99 input channels: two 64-row blocks of that form
])
def test_split_f16_weight_gradient_vs_fp64(eng, forced, cs, cout, grid, kind):
    xs, wf, wg, bias, ref, gy = _ref_and_inputs(cs, cout, grid, kind, 1.0, seed=3 * sum(cs) + cout)
    dev = lambda t: t.detach().float().to(DEV)   # noqa: E731
    grads = []
    for _ in range(2):
        xd = [dev(x).requires_grad_(x.requires_grad) for x in xs]
        wfd, wgd, bd = dev(wf).requires_grad_(True), dev(wg).requires_grad_(True), dev(bias).requires_grad_(True)
        if kind == "gated":
            y = eng.ops.gated_conv3d_act(xd, wfd, wgd, None, bd, act="relu", stride=1)
        else:
            y = eng.ops.conv3d_act(xd, wfd, bd, act="lrelu", stride=1)
        y.backward(dev(gy))
        grads.append((wfd.grad.clone(), wgd.grad.clone() if kind == "gated" else None))
    assert relerr(grads[0][0], wf.grad) < TOL
    if kind == "gated":
        assert relerr(grads[0][1], wg.grad) < TOL
        assert torch.equal(grads[0][1], grads[1][1])
    assert torch.equal(grads[0][0], grads[1][0])          # fixed-order split-K sum: bit-reproducible


# few input channels (csrc/sr3d_hwgrad_fc.hip: Cin <= 5, the MFMA columns are (channel, kx) pairs): conv0's shape (4 features +
# the mask, gated: two dY slices), one channel, three row blocks with a partial last one, several x segments and splits
@pytest.mark.parametrize("cs,cout,grid,kind", [
    ([4, 1], 64, (5, 9, 72), "gated"),
    ([5], 24, (4, 26, 32), "plain"),
    ([3], 130, (4, 12, 40), "plain"),
    ([1], 16, (3, 30, 48), "plain"),
])
def test_split_f16_few_channel_weight_gradient_vs_fp64(eng, forced, cs, cout, grid, kind):
    xs, wf, wg, bias, ref, gy = _ref_and_inputs(cs, cout, grid, kind, 1.0, seed=7 * sum(cs) + cout)
    dev = lambda t: t.detach().float().to(DEV)   # noqa: E731
    grads = []
    for _ in range(2):
        xd = [dev(x).requires_grad_(x.requires_grad) for x in xs]
        wfd, wgd, bd = dev(wf).requires_grad_(True), dev(wg).requires_grad_(True), dev(bias).requires_grad_(True)
        if kind == "gated":
            y = eng.ops.gated_conv3d_act(xd, wfd, wgd, None, bd, act="relu", stride=1)
        else:
            y = eng.ops.conv3d_act(xd, wfd, bd, act="lrelu", stride=1)
        y.backward(dev(gy))
        grads.append((wfd.grad.clone(), wgd.grad.clone() if kind == "gated" else None))
    assert relerr(grads[0][0], wf.grad) < TOL
    if kind == "gated":
        assert relerr(grads[0][1], wg.grad) < TOL
        assert torch.equal(grads[0][1], grads[1][1])
    assert torch.equal(grads[0][0], grads[1][0])          # fixed-order split-K sum: bit-reproducible


def test_split_f16_unshuffle_epilogue_matches_the_fp32_kernel(eng, monkeypatch):
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(2, 33, 5, 7, 33, generator=g) - 0.5).to(DEV)
    w = (torch.randn(72, 33, 3, 3, 3, generator=g) * 0.05).to(DEV)
    b = (torch.randn(72, generator=g) * 0.1).to(DEV)
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("SR3D_SPLIT_F16", mode)
        with torch.no_grad():
            out[mode] = eng.ops.conv3d_act([x], w, b, act="lrelu", unshuffle=True)
    assert out["2"].shape == (2, 9, 10, 14, 66)
    assert relerr(out["2"], out["0"]) < 2e-6


def test_split_f16_has_no_accumulation_bias(eng, forced):
    """The f16 MFMA truncates inside its adder tree: accumulated naively, every output carries a small NEGATIVE error
    (-5e-7 of the output rms at K = 1032) that adds up coherently in sums over voxels (bias gradients).  The kernel
    alternates the sign of accumulator and weights from chunk to chunk; what is left is below 5e-8 of the rms."""
    g = torch.Generator().manual_seed(11)
    for K, N, grid in ((1032, 129, (4, 8, 32)), (257, 264, (2, 4, 32))):
        x = torch.randn(1, K, *grid, generator=g)
        w = torch.randn(N, K, 3, 3, 3, generator=g) * (2.0 / (27 * K)) ** 0.5
        ref = F.conv3d(x.double(), w.double(), None, padding=1)
        with torch.no_grad():
            y = eng.ops.conv3d_act([x.to(DEV)], w.to(DEV), None, act=None, stride=1)
        err = y.cpu().double() - ref
        assert relerr(y, ref) < TOL
        assert abs(err.mean().item()) < 5e-8 * ref.std().item(), (K, err.mean().item() / ref.std().item())


def test_default_mode_takes_the_split_kernel_only_on_chip_filling_launches(eng, monkeypatch):
    """same numbers either way (to the kernels' 1e-6), and the small launch must not depend on the switch being 1 or 0"""
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(1, 64, 4, 8, 32, generator=g) - 0.5).to(DEV)
    w = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).to(DEV)
    out = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("SR3D_SPLIT_F16", mode)
        with torch.no_grad():
            out[mode] = eng.ops.conv3d_act([x], w, None, act=None, stride=1)
    assert torch.equal(out["0"], out["1"])            # 16 workgroups: the Winograd kernel in both
    assert not torch.equal(out["0"], out["2"]) and relerr(out["2"], out["0"]) < 2e-6


def test_unshuffle_layer_accepts_both_row_orders_and_exported_maxima_match_a_sweep(eng, forced):
    """(1) an image packed as SR3D_PACK_FWD (channel order) and one packed as SR3D_PACK_FWD_UNSHUFFLE (rows in voxel-unshuffle
    order, paired stores) give the same output bit for bit; (2) the weight gradient computed from the maxima that the
    forward kernel / the activation-backward kernel export equals the one computed from a sweep of the tensors"""
    import ctypes as C
    from sr3d_amd import _lib as L
    g = torch.Generator().manual_seed(8)
    x = (torch.rand(2, 33, 5, 7, 40, generator=g) - 0.5).to(DEV)
    m = (torch.rand(2, 1, 5, 7, 40, generator=g) > 0.3).float().to(DEV)
    w = (torch.randn(72, 34, 3, 3, 3, generator=g) * 0.05).to(DEV)
    b = (torch.randn(72, generator=g) * 0.1).to(DEV)
    desc = L.conv_desc(2, 34, 72, 5, 7, 40, 1)
    outs = []
    for kind in (L.PACK_FWD, L.PACK_FWD_UNSHUFFLE):
        wp = eng.ops.pack_weights(desc, kind, w, None)
        y = torch.empty(2, 9, 10, 14, 80, device=DEV)
        L.check(L.lib.sr3d_conv3d_fwd(C.byref(desc), L.slices([x, m]), 2, L.dev_ptr(wp), L.dev_ptr(b), L.dev_ptr(y), L.ACT_LRELU, 1,
                                      None, L.stream_ptr()), "fwd")
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    # exported maxima: x slices from the forward kernel, dpre from the LeakyReLU backward
    assert L.lib.sr3d_conv3d_fwd_exports_absmax(C.byref(desc), 0) == 1
    xa = torch.zeros(256, dtype=torch.int32, device=DEV)
    wp = eng.ops.pack_weights(desc, L.PACK_FWD_UNSHUFFLE, w, None)
    L.check(L.lib.sr3d_conv3d_fwd(C.byref(desc), L.slices([x, m]), 2, L.dev_ptr(wp), L.dev_ptr(b), L.dev_ptr(y), L.ACT_LRELU, 1,
                                  C.c_void_p(xa.data_ptr()), L.stream_ptr()), "fwd")
    got = xa.view(torch.float32).view(4, 64).amax(dim=1).cpu()
    # exact per slice, although channels 32 (features), 33 (the mask) share an 8-channel staging group
    assert float(got[0]) == float(x.abs().max()) and float(got[1]) == 1.0
    assert float(got[2]) == 0.0 and float(got[3]) == 0.0
    dy = (torch.rand(2, 72, 5, 7, 40, generator=torch.Generator(device=DEV).manual_seed(1), device=DEV) - 0.5) * 1e-3
    yy = torch.rand(2, 72, 5, 7, 40, generator=torch.Generator(device=DEV).manual_seed(2), device=DEV) - 0.5
    dpre, da = torch.empty_like(dy), torch.zeros(64, dtype=torch.int32, device=DEV)
    L.check(L.lib.sr3d_lrelu_bwd(L.dev_ptr(dy), L.dev_ptr(yy), L.dev_ptr(dpre), dy.numel(), L.DTYPE_F32, C.c_void_p(da.data_ptr()),
                                 L.stream_ptr()), "lrelu_bwd")
    assert float(da.view(torch.float32).max()) == float(dpre.abs().max())
    dw_sweep = eng.ops._bwd_weight(desc, [x, m], [dpre])
    dw_fused = eng.ops._bwd_weight(desc, [x, m], [dpre], xa, da)
    assert relerr(dw_fused, dw_sweep) < 2e-6


def test_stride2_forward_exports_exact_slice_maxima(eng, forced):
    """the stride-2 forward kernel (csrc/sr3d_hconv_s2.hip) exports max|x| per input slice like the stride-1 kernel: exact per
    slice although the mask shares an 8-channel staging group with padding, ragged tiles, batch 2"""
    import ctypes as C
    from sr3d_amd import _lib as L
    g = torch.Generator().manual_seed(9)
    x = ((torch.rand(2, 40, 6, 9, 48, generator=g) - 0.5) * 3.0).to(DEV)
    m = (torch.rand(2, 1, 6, 9, 48, generator=g) > 0.3).float().to(DEV)
    wf = (torch.randn(32, 41, 3, 3, 3, generator=g) * 0.05).to(DEV)
    wg = (torch.randn(32, 41, 3, 3, 3, generator=g) * 0.05).to(DEV)
    bg = (torch.randn(32, generator=g) * 0.1).to(DEV)
    desc = L.conv_desc(2, 41, 32, 6, 9, 48, 2)
    assert L.lib.sr3d_conv3d_fwd_exports_absmax(C.byref(desc), 1) == 1
    wp = eng.ops.pack_weights(desc, L.PACK_FWD_GATED, wf, wg)
    y, ff, ss = (torch.empty(2, 32, 3, 5, 24, device=DEV) for _ in range(3))
    xa = torch.zeros(256, dtype=torch.int32, device=DEV)
    L.check(L.lib.sr3d_gated_conv3d_fwd(C.byref(desc), L.slices([x, m]), 2, L.dev_ptr(wp), None, L.dev_ptr(bg), L.dev_ptr(y),
                                        L.dev_ptr(ff), L.dev_ptr(ss), L.ACT_RELU, C.c_void_p(xa.data_ptr()), L.stream_ptr()), "fwd")
    got = xa.view(torch.float32).view(4, 64).amax(dim=1).cpu()
    assert float(got[0]) == float(x.abs().max()) and float(got[1]) == 1.0
    assert float(got[2]) == 0.0 and float(got[3]) == 0.0


# the stride-2 weight gradient on the f16 MFMA (csrc/sr3d_hwgrad_s2.hip; fine row length a multiple of 16): odd and even
# z / y extents, 64-row and 32-row workgroups, a 2-row last block, a mask slice, 17 channels (a second, almost empty
# 16-channel block), two dY slices (gated), several x segments, batch 2 -- against fp64, and bit-reproducible
@pytest.mark.parametrize("cs,cout,grid,kind", [
    ([64, 1], 64, (8, 16, 64), "gated"),
    ([17], 24, (5, 7, 32), "plain"),
    ([40, 1, 24], 130, (7, 9, 80), "plain"),
    ([33], 40, (6, 12, 48), "gated"),
])
def test_split_f16_stride2_weight_gradient_vs_fp64(eng, forced, cs, cout, grid, kind):
    from sr3d_amd import _lib as L
    g = torch.Generator().manual_seed(17 * sum(cs) + cout)
    B = 2
    xs = [(torch.rand(B, c, *grid, generator=g) - 0.4) if c > 1 else (torch.rand(B, 1, *grid, generator=g) > 0.2).float() for c in cs]
    og = [(n - 1) // 2 + 1 for n in grid]
    n_dy = 2 if kind == "gated" else 1
    dys = [(torch.rand(B, cout, *og, generator=g) - 0.5) * (10.0 ** (-3 * i)) for i in range(n_dy)]
    desc = L.conv_desc(B, sum(cs), cout, *grid, 2)
    outs = [eng.ops._bwd_weight(desc, [x.to(DEV) for x in xs], [d.to(DEV) for d in dys]) for _ in range(2)]
    assert torch.equal(outs[0], outs[1])
    x64 = torch.cat(xs, 1).double()
    for i, d in enumerate(dys):
        wr = torch.zeros(cout, sum(cs), 3, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv3d(x64, wr, None, stride=2, padding=1).backward(d.double())
        assert relerr(outs[0][i * cout:(i + 1) * cout], wr.grad) < TOL, (i, relerr(outs[0][i * cout:(i + 1) * cout], wr.grad))


def test_fused_activation_backward_equals_the_separate_pass_bit_for_bit(eng, monkeypatch):
    """the LeakyReLU layers of the decoder defer their activation backward to the input-gradient epilogue of the
    layer that consumes them (sr3d_conv3d_bwd_data_act, hconv_kernel's plain epilogue).  In fp32 the stored value is the very
    expression of lrelu_bwd_kernel, and the maxima it exports are those of the same values: every parameter gradient of the
    whole model must be BIT-identical with and without the fusion (split kernels forced so that the small grid takes them)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_config, synthetic_batch
    monkeypatch.setenv("SR3D_SPLIT_F16", "2")
    cfg = make_config("mixed")
    x, b, y = synthetic_batch(1, (16, 32, 64), 4, 77, DEV)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(eng.ops, "FUSE_ACT_BWD", fused)
        torch.manual_seed(3)
        model = eng.make_model(cfg).to(DEV)
        calls = {"act": 0, "unsh": 0}
        orig, orig_u = eng._lib.lib.sr3d_lrelu_bwd, eng._lib.lib.sr3d_unshuffle_lrelu_bwd

        def counted(*a, _orig=orig):
            calls["act"] += 1
            return _orig(*a)

        def counted_u(*a, _orig=orig_u):
            calls["unsh"] += 1
            return _orig(*a)
        monkeypatch.setattr(eng._lib.lib, "sr3d_lrelu_bwd", counted)
        monkeypatch.setattr(eng._lib.lib, "sr3d_unshuffle_lrelu_bwd", counted_u)
        loss = eng.make_loss(cfg)(model(x, b), y, b)
        loss.backward()
        torch.cuda.synchronize()
        monkeypatch.setattr(eng._lib.lib, "sr3d_lrelu_bwd", orig)
        monkeypatch.setattr(eng._lib.lib, "sr3d_unshuffle_lrelu_bwd", orig_u)
        res[fused] = ({k: p.grad.detach().clone() for k, p in model.named_parameters()}, (calls["act"], calls["unsh"]), float(loss.detach()))
    assert res[True][2] == res[False][2]
    # 11 plain LeakyReLU layers (3 latent + 8 UpBlock convs) and the 4 unshuffle layers (UpBlock.up): all fused away where the
    # grid allows it (the unshuffle form needs X % 4 == 0 on the consumer's grid: HR x 64, 32, 16, 8 -> all four levels)
    assert res[False][1] == (11, 4) and res[True][1] == (0, 0), (res[True][1], res[False][1])
    for k, g in res[True][0].items():
        assert torch.equal(g, res[False][0][k]), k


def test_skip_gradient_sum_inside_the_gated_backward_equals_autograds_add_bit_for_bit(eng, monkeypatch):
    """The skip tensors f0..f3 (unet.py:262-283) have two consumers; with ``ops.FUSE_SKIP_GRAD_ADD`` the producing gated layer
    returns two handles of its output and adds the two incoming gradients inside sr3d_gated_act_bwd_sum instead of autograd
    adding them in an elementwise pass of its own.  fp32: a + b is the same rounding either way, so every parameter
    gradient must be BIT-identical; the fused form makes 4 two-gradient calls (f0, f1, f2, f3), the other none."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_config, synthetic_batch
    cfg = make_config("mixed")
    x, b, y = synthetic_batch(2, (16, 32, 64), 4, 78, DEV)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(eng.ops, "FUSE_SKIP_GRAD_ADD", fused)
        torch.manual_seed(5)
        model = eng.make_model(cfg).to(DEV)
        two = {"n": 0, "calls": 0}
        orig = eng._lib.lib.sr3d_gated_act_bwd_sum

        def counted(dy, dy2, *a, _orig=orig):
            two["calls"] += 1
            two["n"] += 1 if (dy2 is not None and getattr(dy2, "value", None)) else 0
            return _orig(dy, dy2, *a)
        monkeypatch.setattr(eng._lib.lib, "sr3d_gated_act_bwd_sum", counted)
        loss = eng.make_loss(cfg)(model(x, b), y, b)
        loss.backward()
        torch.cuda.synchronize()
        monkeypatch.setattr(eng._lib.lib, "sr3d_gated_act_bwd_sum", orig)
        res[fused] = ({k: p.grad.detach().clone() for k, p in model.named_parameters()}, (two["calls"], two["n"]), float(loss.detach()))
    assert res[True][2] == res[False][2]
    assert res[True][1] == (9, 4) and res[False][1] == (9, 0), (res[True][1], res[False][1])     # 9 gated layers, 4 skip tensors
    for k, g in res[True][0].items():
        assert torch.equal(g, res[False][0][k]), k

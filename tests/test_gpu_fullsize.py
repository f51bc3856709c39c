"""Parity at BASELINE.json's full grid (HR 80x320x320) through size-independent
properties: locality (a crop of the full-size result equals the oracle run on
the cropped input), linearity / support of the gradients, determinism.  The CPU
oracle only ever sees small crops, so this stays within seconds."""
import pytest
import torch
import torch.nn.functional as F

from helpers import relerr
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FULL = (80, 320, 320)
TOL = 1e-5


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available()
    import sr3d_amd
    return sr3d_amd


def _crop(t, lo, hi):
    return t[:, :, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]


@pytest.mark.parametrize("stride,gated", [(1, True), (2, True), (1, False)])
def test_forward_locality_at_full_size(eng, stride, gated):
    """conv at 80x320x320 vs oracle on crops that touch a corner, an edge and the interior"""
    g = torch.Generator().manual_seed(11 + stride)
    cin_a, cout = 12, 40
    xa = torch.rand(1, cin_a, *FULL, generator=g) - 0.5
    xb = (torch.rand(1, 1, *FULL, generator=g) > 0.2).float()          # mask slice of the virtual concat
    wf = torch.randn(cout, cin_a + 1, 3, 3, 3, generator=g) * 0.1
    wg = torch.randn(cout, cin_a + 1, 3, 3, 3, generator=g) * 0.1
    bg = torch.randn(cout, generator=g) * 0.1
    if gated:
        y = eng.ops.gated_conv3d_act([xa.to(DEV), xb.to(DEV)], wf.to(DEV), wg.to(DEV), None, bg.to(DEV), act="relu",
                                     stride=stride).cpu()
    else:
        y = eng.ops.conv3d_act([xa.to(DEV), xb.to(DEV)], wf.to(DEV), bg.to(DEV), act="lrelu", stride=stride).cpu()
    x = torch.cat([xa, xb], 1)
    # output windows [lo, hi) on the OUTPUT grid
    for lo, hi in [((0, 0, 0), (6, 10, 40)), ((70 // stride, 300 // stride, 280 // stride), tuple(f // stride for f in FULL)),
                   ((31 // stride, 157 // stride, 95 // stride), (31 // stride + 5, 157 // stride + 7, 95 // stride + 33))]:
        ilo = [max(0, l * stride - 1) for l in lo]
        ihi = [min(f, (h - 1) * stride + 2) for h, f in zip(hi, FULL)]
        xc = _crop(x, ilo, ihi)
        # pad so that the crop's own zero padding coincides with the true border only where it is a true border
        pad = []
        for d in (2, 1, 0):
            pad += [1 if lo[d] * stride - 1 < 0 else 0, 1 if (hi[d] - 1) * stride + 2 > FULL[d] else 0]
        xc = F.pad(xc, pad)
        if gated:
            ref = torch.sigmoid(F.conv3d(xc, wg, bg, stride=stride)) * F.relu(F.conv3d(xc, wf, None, stride=stride))
        else:
            ref = F.leaky_relu(F.conv3d(xc, wf, bg, stride=stride), 0.01)
        got = _crop(y, lo, hi)
        assert got.shape == ref.shape, (got.shape, ref.shape)
        assert relerr(got, ref) < TOL


def test_backward_support_and_values_at_full_size(eng):
    """dy is non-zero only in a window: dx must vanish outside its 1-voxel dilation and, like dW, equal the
    oracle evaluated on the cropped problem (both gradients are linear in dy)"""
    g = torch.Generator().manual_seed(5)
    cin, cout = 20, 33
    x = (torch.rand(1, cin, *FULL, generator=g) - 0.5)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.1)
    lo, hi = (40, 100, 200), (46, 109, 233)
    dy = torch.zeros(1, cout, *FULL)
    dy[:, :, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = torch.rand(1, cout, 6, 9, 33, generator=g) - 0.5
    xd = x.to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    y = eng.ops.conv3d_act([xd], wd, None, act=None)
    y.backward(dy.to(DEV))
    dx, dw = xd.grad.cpu(), wd.grad.cpu()
    ilo, ihi = [l - 1 for l in lo], [h + 1 for h in hi]
    outside = dx.clone()
    outside[:, :, ilo[0]:ihi[0], ilo[1]:ihi[1], ilo[2]:ihi[2]] = 0
    assert float(outside.abs().max()) == 0.0
    # oracle on the crop (window + halo), no padding needed: the window is interior
    xc = _crop(x, ilo, ihi).clone().requires_grad_(True)
    wc = w.clone().requires_grad_(True)
    F.conv3d(xc, wc, None).backward(_crop(dy, lo, hi))
    assert relerr(_crop(dx, ilo, ihi), xc.grad) < TOL
    assert relerr(dw, wc.grad) < TOL


def test_unshuffle_conv_full_level1(eng):
    """UpBlock.up at level 1 (40x160x160 -> 80x320x320): crop of the scattered output vs oracle"""
    g = torch.Generator().manual_seed(9)
    cin = 9
    x = torch.rand(1, cin, 40, 160, 160, generator=g) - 0.5
    w = torch.randn(8 * cin, cin, 3, 3, 3, generator=g) * 0.1
    b = torch.randn(8 * cin, generator=g) * 0.1
    y = eng.ops.conv3d_act([x.to(DEV)], w.to(DEV), b.to(DEV), act="lrelu", unshuffle=True).cpu()
    assert tuple(y.shape) == (1, cin, 80, 320, 320)
    lo, hi = (17, 60, 96), (23, 66, 130)     # coarse window, interior
    xc = _crop(x, [l - 1 for l in lo], [h + 1 for h in hi])
    ref = R.unshuffle_voxels(F.leaky_relu(F.conv3d(xc, w, b), 0.01), 2)
    got = _crop(y, [2 * l for l in lo], [2 * h for h in hi])
    assert relerr(got, ref) < TOL


@pytest.mark.parametrize("B", [1, 4])
def test_losses_at_full_size(eng, B):
    """B = 4 is the batch of BASELINE configs[2] (80x320x320, physics loss)"""
    g = torch.Generator().manual_seed(3)
    p = torch.rand(B, 4, *FULL, generator=g)
    t = torch.rand(B, 4, *FULL, generator=g)
    b = (torch.rand(B, 1, *FULL, generator=g) > 0.2).float()
    scales = [14.4, 21.6, 7.0]
    pd = p.to(DEV).requires_grad_(True)
    terms = eng.ops.MixedLossFn.apply(pd, t.to(DEV), b.to(DEV), scales, 5.0, 1.0, 10.0)
    terms[3].backward()
    pr = p.clone().requires_grad_(True)
    ref_terms = R.mixed_div_grad_terms(pr, t, b, 1.0, 10.0, scales)
    ref_total = ref_terms[0] + 1.0 * ref_terms[1] + 10.0 * ref_terms[2]
    ref_total.backward()
    for a, r in zip(terms[:3].tolist(), ref_terms):
        assert abs(a - float(r.detach())) < TOL * abs(float(r.detach()))
    assert abs(float(terms[3]) - float(ref_total)) < TOL * float(ref_total)
    assert relerr(pd.grad, pr.grad) < TOL
    pd2 = p.to(DEV).requires_grad_(True)
    l1 = eng.ops.L1LossFn.apply(pd2, t.to(DEV))
    l1.backward()
    assert abs(float(l1) - float((p - t).abs().mean())) < TOL * float(l1)
    assert relerr(pd2.grad, torch.sign(p - t) / p.numel()) < 1e-6


def test_batch4_gated_stride2_full_size(eng):
    """down1.convs.0 of BASELINE configs[2]: gated 65 -> 128, stride 2, batch 4 at 80x320x320 (virtual concat of a
    64-channel tensor and the mask).  Windows of the LAST sample against the oracle on the cropped input."""
    g = torch.Generator(device=DEV).manual_seed(21)
    B, ca, cout = 4, 64, 128
    xa = torch.rand(B, ca, *FULL, generator=g, device=DEV) - 0.5
    xb = (torch.rand(B, 1, *FULL, generator=g, device=DEV) > 0.2).float()
    wf = torch.randn(cout, ca + 1, 3, 3, 3, generator=g, device=DEV) * 0.05
    wg = torch.randn(cout, ca + 1, 3, 3, 3, generator=g, device=DEV) * 0.05
    bg = torch.randn(cout, generator=g, device=DEV) * 0.1
    y = eng.ops.gated_conv3d_act([xa, xb], wf, wg, None, bg, act="relu", stride=2)
    assert tuple(y.shape) == (B, cout, 40, 160, 160)
    wfc, wgc, bgc = wf.cpu(), wg.cpu(), bg.cpu()
    for smp in (0, B - 1):
        for lo, hi in [((0, 0, 0), (3, 5, 20)), ((36, 150, 128), (40, 160, 160)), ((17, 77, 45), (20, 82, 78))]:
            ilo = [max(0, 2 * l - 1) for l in lo]
            ihi = [min(f, 2 * (h - 1) + 2) for h, f in zip(hi, FULL)]
            xc = torch.cat([_crop(xa[smp:smp + 1], ilo, ihi), _crop(xb[smp:smp + 1], ilo, ihi)], 1).cpu()
            pad = []
            for dd in (2, 1, 0):
                pad += [1 if 2 * lo[dd] - 1 < 0 else 0, 1 if 2 * (hi[dd] - 1) + 2 > FULL[dd] else 0]
            xc = F.pad(xc, pad)
            ref = torch.sigmoid(F.conv3d(xc, wgc, bgc, stride=2)) * F.relu(F.conv3d(xc, wfc, None, stride=2))
            got = _crop(y[smp:smp + 1], lo, hi).cpu()
            assert got.shape == ref.shape
            assert relerr(got, ref) < TOL, (smp, lo)


def test_batch4_unshuffle_conv_beyond_2g_elements(eng):
    """up1.up.0 of BASELINE configs[2]: 129 -> 1032 + bias + LeakyReLU + voxel unshuffle at batch 4; the output
    (4, 129, 80, 320, 320) has 4.2e9 elements, so the last sample sits beyond 2^31 ELEMENTS (not just bytes).
    Forward windows, and -- with dy supported in a window of the last sample -- input and weight gradients."""
    g = torch.Generator(device=DEV).manual_seed(22)
    B, ca = 4, 128
    lr_grid = (40, 160, 160)
    xa = (torch.rand(B, ca, *lr_grid, generator=g, device=DEV) - 0.5).requires_grad_(True)
    xb = (torch.rand(B, 1, *lr_grid, generator=g, device=DEV) > 0.2).float()
    w = (torch.randn(8 * (ca + 1), ca + 1, 3, 3, 3, generator=g, device=DEV) * 0.03).requires_grad_(True)
    bias = (torch.randn(8 * (ca + 1), generator=g, device=DEV) * 0.1).requires_grad_(True)
    y = eng.ops.conv3d_act([xa, xb], w, bias, act="lrelu", unshuffle=True)
    assert tuple(y.shape) == (B, ca + 1, *FULL) and y.numel() > 2 ** 31
    wc, bc = w.detach().cpu(), bias.detach().cpu()
    smp = B - 1
    lo, hi = (30, 150, 120), (34, 156, 154)                  # coarse window of the last sample, interior
    ilo, ihi = [l - 1 for l in lo], [h + 1 for h in hi]
    xc = torch.cat([_crop(xa[smp:smp + 1].detach(), ilo, ihi), _crop(xb[smp:smp + 1], ilo, ihi)], 1).cpu()
    xcr, wcr, bcr = xc.clone().requires_grad_(True), wc.clone().requires_grad_(True), bc.clone().requires_grad_(True)
    pre = F.conv3d(xcr, wcr, bcr)
    ref = R.unshuffle_voxels(F.leaky_relu(pre, 0.01), 2)
    got = _crop(y[smp:smp + 1].detach(), [2 * l for l in lo], [2 * h for h in hi]).cpu()
    assert relerr(got, ref) < TOL
    # also the first voxels of sample 0 and the last voxels of the last sample (the far end of the 17 GB tensor)
    for s2, lo2, hi2 in [(0, (0, 0, 0), (2, 3, 18)), (smp, (38, 157, 140), (40, 160, 160))]:
        i0 = [max(0, l - 1) for l in lo2]
        i1 = [min(f, h + 1) for h, f in zip(hi2, lr_grid)]
        x2 = torch.cat([_crop(xa[s2:s2 + 1].detach(), i0, i1), _crop(xb[s2:s2 + 1], i0, i1)], 1).cpu()
        pad = []
        for dd in (2, 1, 0):
            pad += [1 if lo2[dd] - 1 < 0 else 0, 1 if hi2[dd] + 1 > lr_grid[dd] else 0]
        r2 = R.unshuffle_voxels(F.leaky_relu(F.conv3d(F.pad(x2, pad), wc, bc), 0.01), 2)
        g2 = _crop(y[s2:s2 + 1].detach(), [2 * l for l in lo2], [2 * h for h in hi2]).cpu()
        assert relerr(g2, r2) < TOL, (s2, lo2)

    # backward: dy lives in the fine-grid window of the last sample only, and not on elements next to the kink
    dyc = (torch.rand(ref.shape) - 0.5) * R.unshuffle_voxels((pre.detach().abs() > 1e-5).float(), 2)
    ref.backward(dyc)
    dy = torch.zeros_like(y)
    dy[smp:smp + 1, :, 2 * lo[0]:2 * hi[0], 2 * lo[1]:2 * hi[1], 2 * lo[2]:2 * hi[2]] = dyc.to(DEV)
    y.backward(dy)
    del dy, y
    dx = xa.grad
    assert float(dx[:smp].abs().max()) == 0.0
    outside = dx[smp:smp + 1].clone()
    outside[:, :, ilo[0]:ihi[0], ilo[1]:ihi[1], ilo[2]:ihi[2]] = 0
    assert float(outside.abs().max()) == 0.0
    assert relerr(_crop(dx[smp:smp + 1], ilo, ihi), xcr.grad[:, :ca]) < TOL
    assert relerr(w.grad, wcr.grad) < TOL
    assert relerr(bias.grad, bcr.grad) < TOL


@pytest.mark.parametrize("name,cs,cout,gated,grid", [
    ("up1.convs.0", [64, 1, 129], 64, False, FULL),             # 194 -> 64 over three slices, LeakyReLU
    ("down1.convs.1", [128], 128, True, (40, 160, 160)),        # gated 128 -> 128, ReLU
])
def test_hconv_launch_shapes_vs_oracle_windows_at_full_size(eng, name, cs, cout, gated, grid):
    """hconv_kernel at BASELINE size against the ORACLE (not against another HIP engine): the two heaviest launch
    shapes besides up1.up.0 -- forward windows (corner, far corner, interior), then, with dy supported in an interior
    window, the input gradient of every slice (zero outside the window's 1-voxel dilation, equal to the oracle's on the
    cropped problem inside), the weight gradient and the bias gradient.  reference: unet.py:72-97, custom_conv.py:111-126"""
    g = torch.Generator(device=DEV).manual_seed(41 + cout)
    cin = sum(cs)
    xs = [((torch.rand(1, c, *grid, generator=g, device=DEV) - 0.5) if c > 1 else
           (torch.rand(1, 1, *grid, generator=g, device=DEV) > 0.2).float()).requires_grad_(c > 1) for c in cs]
    std = (2.0 / (27 * cin)) ** 0.5
    wf = (torch.randn(cout, cin, 3, 3, 3, generator=g, device=DEV) * std).requires_grad_(True)
    wg = (torch.randn(cout, cin, 3, 3, 3, generator=g, device=DEV) * std).requires_grad_(True)
    bias = (torch.randn(cout, generator=g, device=DEV) * 0.1).requires_grad_(True)
    if gated:
        y = eng.ops.gated_conv3d_act(xs, wf, wg, None, bias, act="relu", stride=1)
    else:
        y = eng.ops.conv3d_act(xs, wf, bias, act="lrelu", stride=1)
    wfc, wgc, bc = wf.detach().cpu(), wg.detach().cpu(), bias.detach().cpu()

    def oracle(xc, wf_, wg_, b_):
        if gated:
            pre = F.conv3d(xc, wf_, None)
            return torch.sigmoid(F.conv3d(xc, wg_, b_)) * F.relu(pre), pre
        pre = F.conv3d(xc, wf_, b_)
        return F.leaky_relu(pre, 0.01), pre

    def crop_in(lo, hi):
        ilo, ihi = [max(0, l - 1) for l in lo], [min(f, h + 1) for h, f in zip(hi, grid)]
        xc = torch.cat([_crop(x.detach(), ilo, ihi) for x in xs], 1).cpu()
        pad = []
        for dd in (2, 1, 0):
            pad += [1 if lo[dd] - 1 < 0 else 0, 1 if hi[dd] + 1 > grid[dd] else 0]
        return F.pad(xc, pad), ilo, ihi

    far = tuple(f - d for f, d in zip(grid, (3, 6, 34)))
    mid = (grid[0] // 2 - 3, grid[1] // 2 + 1, grid[2] // 2 - 17)
    for lo, hi in [((0, 0, 0), (3, 5, 36)), (far, grid), (mid, tuple(m + d for m, d in zip(mid, (5, 6, 35))))]:
        xc, _, _ = crop_in(lo, hi)
        ref, _ = oracle(xc, wfc, wgc, bc)
        assert relerr(_crop(y.detach(), lo, hi).cpu(), ref) < TOL, (name, lo)

    lo, hi = mid, tuple(m + d for m, d in zip(mid, (5, 6, 35)))
    xc, ilo, ihi = crop_in(lo, hi)
    xcr, wfr, wgr, br = (t.clone().requires_grad_(True) for t in (xc, wfc, wgc, bc))
    ref, pre = oracle(xcr, wfr, wgr, br)
    dyc = (torch.rand(ref.shape) - 0.5) * (pre.detach().abs() > 1e-5).float()    # no gradient through the kink
    ref.backward(dyc)
    dy = torch.zeros_like(y)
    dy[:, :, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = dyc.to(DEV)
    y.backward(dy)
    c0 = 0
    for x, c in zip(xs, cs):
        if x.requires_grad:
            outside = x.grad.clone()
            outside[:, :, ilo[0]:ihi[0], ilo[1]:ihi[1], ilo[2]:ihi[2]] = 0
            assert float(outside.abs().max()) == 0.0, (name, c)
            assert relerr(_crop(x.grad, ilo, ihi), xcr.grad[:, c0:c0 + c]) < TOL, (name, c)
        c0 += c
    assert relerr(wf.grad, wfr.grad) < TOL and relerr(bias.grad, br.grad) < TOL, name
    if gated:
        assert relerr(wg.grad, wgr.grad) < TOL, name


@pytest.mark.parametrize("cs,cout,stride,gated,B,grid", [
    ([64], 64, 1, False, 1, FULL),            # up1.convs.1: 80 workgroup rounds, 10 x segments, 26 splits
    ([64, 1], 128, 2, True, 1, FULL),         # down1.convs.0 (stride 2, gated, mask slice)
    ([128, 1], 136, 1, False, 4, (40, 160, 160)),   # batch 4 at level 1: sample offsets, a 2-row last block, few-channel kernel
])
def test_split_f16_and_fp32_engines_agree_at_full_size(eng, monkeypatch, cs, cout, stride, gated, B, grid):
    """the split-f16 kernels (forward, input gradient, weight gradient; default dispatch) against the fp32 MFMA kernels
    (SR3D_SPLIT_F16=0) on the benchmark's own launch shapes; no activation kinks, so both are the same smooth function"""
    g = torch.Generator().manual_seed(31 + cout)
    xs = [(torch.rand(B, c, *grid, generator=g) - 0.5) if c > 1 else (torch.rand(B, 1, *grid, generator=g) > 0.2).float()
          for c in cs]
    cin = sum(cs)
    wf = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    wg = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    out = {}
    gy = None
    for mode in ("0", "1"):
        monkeypatch.setenv("SR3D_SPLIT_F16", mode)
        xd = [x.to(DEV).requires_grad_(x.shape[1] > 1) for x in xs]
        wfd, wgd, bd = (t.to(DEV).requires_grad_(True) for t in (wf, wg, bias))
        if gated:
            y = eng.ops.gated_conv3d_act(xd, wfd, wgd, None, bd, act=None, stride=stride)
        else:
            y = eng.ops.conv3d_act(xd, wfd, bd, act=None, stride=stride)
        if gy is None:
            gy = torch.rand(y.shape, generator=torch.Generator(device=DEV).manual_seed(5), device=DEV) - 0.5
        y.backward(gy)
        torch.cuda.synchronize()
        out[mode] = [y.detach(), xd[0].grad, wfd.grad, bd.grad] + ([wgd.grad] if gated else [])
        del xd, y
    for a, b in zip(out["1"], out["0"]):
        assert relerr(a, b) < 5e-6
    assert not torch.equal(out["1"][0], out["0"][0])     # (two different engines did run)


def test_fused_adam_65m_parameters(eng):
    n = 65_472_736
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(n, generator=g) * 0.05
    grads = [torch.randn(n, generator=g) * 1e-3 for _ in range(2)]
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref_p], lr=1e-4)
    pd = p0.to(DEV)
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    for step, gr in enumerate(grads, 1):
        ref_p.grad = gr.clone()
        opt.step()
        eng.ops.adam_step_(pd, gr.to(DEV), m, v, 1e-4, 0.9, 0.999, 1e-8, step)
    assert relerr(pd.cpu() - p0, ref_p.detach() - p0) < 1e-5    # the UPDATE, not just the weights
    assert relerr(m.cpu(), opt.state[ref_p]["exp_avg"]) < 1e-6
    assert relerr(v.cpu(), opt.state[ref_p]["exp_avg_sq"]) < 1e-6


def test_training_step_is_bit_reproducible(eng):
    """two identical steps from the same state give identical gradients (deterministic wgrad / reductions);
    default.yml widths on a reduced grid to keep the test short"""
    import bench
    cfg = bench.make_config("mixed")
    torch.manual_seed(0)
    model = eng.make_model(cfg).to(DEV)
    loss_fn = eng.make_loss(cfg)
    x, b, y = bench.synthetic_batch(1, (16, 64, 64), 4, 7, DEV)
    outs = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model(x, b), y, b)
        loss.backward()
        outs.append((float(loss), torch.cat([p.grad.flatten() for p in model.parameters()]).clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[1] as a WHOLE: one training step of the default.yml model at HR 80x320x320, checked layer by layer
# ------------------------------------------------------------------------------------------------------------------
def _window_oracle(call, lo, hi):
    """the oracle's output of one recorded layer on the output window [lo, hi), from a crop of the layer's REAL inputs"""
    s = call["stride"]
    grid = tuple(call["srcs"][0].shape[2:])
    ilo = [max(0, l * s - 1) for l in lo]
    ihi = [min(f, (h - 1) * s + 2) for h, f in zip(hi, grid)]
    xc = torch.cat([_crop(t, ilo, ihi) for t in call["srcs"]], 1).float().cpu()
    pad = []
    for d in (2, 1, 0):
        pad += [1 if lo[d] * s - 1 < 0 else 0, 1 if (hi[d] - 1) * s + 2 > grid[d] else 0]
    xc = F.pad(xc, pad)
    w = call["w"].detach().cpu()
    bias = call["bias"].detach().cpu() if call["bias"] is not None else None
    if call["kind"] == "gated":
        feat = F.conv3d(xc, w, call["b_feat"].detach().cpu() if call["b_feat"] is not None else None, stride=s)
        gate = torch.sigmoid(F.conv3d(xc, call["w_gate"].detach().cpu(), bias, stride=s))
        return gate * (F.relu(feat) if call["act"] == "relu" else feat)
    pre = F.conv3d(xc, w, bias, stride=s)
    out = F.leaky_relu(pre, 0.01) if call["act"] == "lrelu" else pre
    return R.unshuffle_voxels(out, 2) if call["unshuffle"] else out


def test_whole_model_step_at_baseline_size_layer_by_layer_vs_oracle(eng, monkeypatch):
    """One training step (forward, L1 loss, backward) of the default.yml model on BASELINE configs[1]'s volume -- LR 20x80x80
    -> HR 80x320x320, the bench's own seeds and inputs, default dispatch -- with every convolution checked IN PLACE against
    the oracle: the U-Net's receptive field (~170 voxels) rules out cropping the model, so each of its 34 layers is
    compared on windows (corner, far corner, interior) with the oracle applied to crops of the inputs that layer really
    received; for the layers that carry the step's time (`last`, up1.convs.1, up1.convs.0, up1.up.0) the input gradient is
    checked the same way on windows, and the weight and bias gradients -- sums over all 8.2 M voxels -- against the
    oracle's tap-by-tap contraction evaluated in fp64 (oracle/ref_cpu.py:conv3d_weight_grad_by_taps, pinned on the CPU).
    reference: pytorch/model/unet.py:253-297, pytorch/src/optim_helper.py:160-165."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_config, synthetic_batch
    cfg = make_config("l1")
    torch.manual_seed(42)
    model = eng.make_model(cfg).to(DEV)
    names = {id(p): n for n, p in model.named_parameters()}
    x, b, y = synthetic_batch(1, FULL, 4, 1234, DEV)
    calls = []
    orig_plain, orig_gated = eng.ops.conv3d_act, eng.ops.gated_conv3d_act

    def rec_plain(srcs, weight, bias=None, act=None, stride=1, unshuffle=False, defer_act_bwd=False, out_fp32=False):
        out = orig_plain(srcs, weight, bias, act=act, stride=stride, unshuffle=unshuffle, defer_act_bwd=defer_act_bwd, out_fp32=out_fp32)
        calls.append({"kind": "plain", "name": names[id(weight)].rsplit(".", 1)[0], "srcs": [t.detach() for t in srcs],
                      "src_objs": list(srcs), "w": weight, "bias": bias, "act": act, "stride": stride, "unshuffle": unshuffle,
                      "out": out, "deferred": getattr(out, "_sr3d_act_box", None) is not None})
        return out

    def rec_gated(srcs, w_feat, w_gate, b_feat, b_gate, act=None, stride=1, dual=False):
        out = orig_gated(srcs, w_feat, w_gate, b_feat, b_gate, act=act, stride=stride, dual=dual)
        # (dual: the skip tensors leave as two handles of one storage; the first is the layer's output for the checks below)
        calls.append({"kind": "gated", "name": names[id(w_feat)].rsplit(".", 1)[0], "srcs": [t.detach() for t in srcs],
                      "w": w_feat, "w_gate": w_gate, "b_feat": b_feat, "bias": b_gate, "act": act, "stride": stride,
                      "unshuffle": False, "out": out[0] if dual else out})
        return out

    monkeypatch.setattr(eng.ops, "conv3d_act", rec_plain)
    monkeypatch.setattr(eng.ops, "gated_conv3d_act", rec_gated)
    pred = model(x, b)
    monkeypatch.undo()
    assert len(calls) == 34 - 9           # 34 nn.Conv3d of the reference = 9 fused gated pairs + 16 plain layers
    assert tuple(pred.shape) == (1, 4, *FULL)

    # ---- forward: every layer on three windows of its own output grid
    worst = ("", 0.0)
    for c in calls:
        og = tuple(c["out"].shape[2:])
        f = 2 if c["unshuffle"] else 1                  # windows are chosen on the conv's (coarse) output grid
        cg = tuple(v // f for v in og)
        size = (3, 5, 36)
        far = tuple(max(0, g - d) for g, d in zip(cg, size))
        mid = tuple(max(0, min(g - d, g // 2 - d // 2 + o)) for g, d, o in zip(cg, size, (1, 1, -3)))
        for lo in {(0, 0, 0), far, mid}:
            hi = tuple(min(g, l + d) for g, l, d in zip(cg, lo, size))
            ref = _window_oracle(c, lo, hi)
            got = _crop(c["out"].detach(), [f * l for l in lo], [f * h for h in hi]).float().cpu()
            e = relerr(got, ref)
            worst = max(worst, (c["name"], e), key=lambda t: t[1])
            assert e < TOL, (c["name"], lo, e)
    print(f"forward, 25 fused layers x 3 windows at full size: worst {worst[0]} {worst[1]:.2e}")

    # ---- loss and backward
    by_name = {c["name"]: c for c in calls}
    heavy = ["last", "up1.convs.1.conv", "up1.convs.0.conv", "up1.up.0"]
    dys = {}
    for n in heavy + ["up2.convs.1.conv"]:          # (the last one: its gradient is up1.up.0's input gradient)
        by_name[n]["out"].register_hook(lambda g, n=n: dys.__setitem__(n, g.detach()))
    loss = eng.make_loss(cfg)(pred, y, b)
    ref_loss = (pred.detach().double() - y.double()).abs().mean()
    assert abs(float(loss.detach()) - float(ref_loss)) <= TOL * float(ref_loss)
    model.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
    assert relerr(dys["last"], torch.sign(pred.detach() - y) / pred.numel()) < 1e-6       # dL/dpred of the L1 loss
    # up1.convs.0 / .1 deferred their activation backward: the gradient that reaches them through autograd is
    # already dL/dpre, written by the consumer's input-gradient epilogue; the window check below holds it against
    # conv_transpose(...) * lrelu'(y) of the oracle, the weight gradients against the tap contraction of that dL/dpre
    assert by_name["up1.convs.1.conv"]["deferred"] and by_name["up1.convs.0.conv"]["deferred"] and by_name["up1.up.0"]["deferred"]
    for n in heavy:
        c = by_name[n]
        out, dy = c["out"].detach(), dys[n]
        if c["act"] == "lrelu" and not c["deferred"]:
            dpre = dy * torch.where(out > 0, 1.0, 0.01)
            # (an element within rounding distance of the kink may fall on either side; none of these layers' outputs is
            #  exactly 0 on this data, and the engine decides on the same stored y)
        else:
            dpre = dy
        if c["unshuffle"] and c["deferred"]:
            # the consumer wrote up1.up.0's dL/dpre in THAT layer's layout (8 C channels on the coarse grid) into a buffer that
            # autograd sees with the unshuffled shape: same bytes, viewed back
            B_, c_, z2, y2, x2 = dy.shape
            dpre = dy.contiguous().view(B_, 8 * c_, z2 // 2, y2 // 2, x2 // 2)
        elif c["unshuffle"]:
            dpre = R.shuffle_voxels(dpre, 2)
        xin = torch.cat(c["srcs"], 1)
        dw_ref, db_ref = R.conv3d_weight_grad_by_taps(xin, dpre)
        e_w = relerr(c["w"].grad, dw_ref)
        print(f"{n}: weight gradient vs fp64 tap contraction over {xin[0, 0].numel()} voxels: {e_w:.2e}")
        assert e_w < TOL, (n, e_w)
        if c["bias"] is not None:
            assert relerr(c["bias"].grad, db_ref) < TOL, n
        del xin, dw_ref
        # input gradient of the slices that carry one, on windows: conv_transpose of the (1-dilated) dpre crop
        grid = tuple(c["srcs"][0].shape[2:])
        w_cpu = c["w"].detach().cpu()
        c0 = 0
        for src, obj in zip(c["srcs"], c["src_objs"]):
            ch = src.shape[1]
            prod = next((k for k in calls if k["out"] is obj), None)
            if prod is not None and prod["name"] in dys and obj.requires_grad:
                got_full = dys[prod["name"]]            # the gradient that reached the producer = this layer's dx
                if prod["deferred"] and prod["unshuffle"]:      # (stored in the producer's shuffled layout: back to the fine grid)
                    B_, c_, z2, y2, x2 = got_full.shape
                    got_full = R.unshuffle_voxels(got_full.contiguous().view(B_, 8 * c_, z2 // 2, y2 // 2, x2 // 2), 2)
                for lo in [(0, 0, 0), tuple(g - d for g, d in zip(grid, (3, 5, 36))), (grid[0] // 2, grid[1] // 2 + 1, grid[2] // 2 - 9)]:
                    hi = tuple(min(g, l + d) for g, l, d in zip(grid, lo, (3, 5, 36)))
                    dlo, dhi = [max(0, l - 1) for l in lo], [min(g, h + 1) for h, g in zip(hi, grid)]
                    dcrop = _crop(dpre, dlo, dhi).float().cpu()
                    full = F.conv_transpose3d(dcrop, w_cpu[:, c0:c0 + ch], None, padding=1)     # grid [dlo, dhi)
                    ref = _crop(full, [l - d for l, d in zip(lo, dlo)], [h - d for h, d in zip(hi, dlo)])
                    if prod["deferred"]:        # what arrived there is dL/dy * lrelu'(y): the fused epilogue
                        ref = ref * torch.where(_crop(prod["out"].detach(), lo, hi).float().cpu() > 0, 1.0, 0.01)
                    e = relerr(_crop(got_full, lo, hi).float().cpu(), ref)
                    assert e < TOL, (n, "dx", prod["name"], lo, e)
            c0 += ch
        del dpre


def test_bf16_unshuffle_layer_at_configs4_size_beyond_2_32_elements_per_sample(eng):
    """BASELINE configs[4]'s heaviest layer at ITS size: up1.up.0, 129 -> 1032 + bias + LeakyReLU + voxel unshuffle, bf16
    storage, ONE sample on the 80x320x320 coarse grid -> (1, 129, 160, 640, 640) = 8.45e9 elements (17 GB): per-sample
    element offsets beyond 2^32.  Operands are bf16-representable, so products are exact and the fp64 oracle on crops
    is the arbiter (tests/test_gpu_bf16.py): forward windows at both ends of the tensor and in the interior; with dy
    supported in a far window, the input gradient (zero elsewhere), the weight and the bias gradient."""
    BF = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(404)
    ca = 128
    cgrid = (80, 320, 320)
    xa = (torch.rand(1, ca, *cgrid, generator=g, device=DEV) - 0.3).to(BF).requires_grad_(True)
    xb = (torch.rand(1, 1, *cgrid, generator=g, device=DEV) > 0.2).to(BF)
    w = (torch.randn(8 * (ca + 1), ca + 1, 3, 3, 3, generator=g, device=DEV) * 0.03).to(BF).float().requires_grad_(True)
    bias = (torch.randn(8 * (ca + 1), generator=g, device=DEV) * 0.1).requires_grad_(True)
    y = eng.ops.conv3d_act([xa, xb], w, bias, act="lrelu", unshuffle=True)
    assert y.dtype == BF and tuple(y.shape) == (1, ca + 1, 160, 640, 640) and y[0].numel() > 2 ** 32
    w64, b64 = w.detach().double().cpu(), bias.detach().double().cpu()

    def q(t):
        return t.float().to(BF).double()

    def one_ulp(got, ref, what):
        got = got.float().cpu().double()
        tol = 2.0 ** -7 * ref.abs() + 1e-5 * float(ref.pow(2).mean().sqrt())
        bad = int(((got - ref).abs() > tol).sum())
        assert bad == 0, (what, bad)

    far_lo = (76, 313, 284)
    windows = [((0, 0, 0), (3, 5, 36)), (far_lo, cgrid), ((41, 158, 150), (44, 163, 186))]
    for lo, hi in windows:
        i0 = [max(0, l - 1) for l in lo]
        i1 = [min(f, h + 1) for h, f in zip(hi, cgrid)]
        xc = torch.cat([_crop(xa.detach(), i0, i1), _crop(xb, i0, i1)], 1).double().cpu()
        pad = []
        for dd in (2, 1, 0):
            pad += [1 if lo[dd] - 1 < 0 else 0, 1 if hi[dd] + 1 > cgrid[dd] else 0]
        ref = q(R.unshuffle_voxels(F.leaky_relu(F.conv3d(F.pad(xc, pad), w64, b64), 0.01), 2))
        one_ulp(_crop(y.detach(), [2 * l for l in lo], [2 * h for h in hi]), ref, ("y", lo))

    # backward: dy supported in the FAR window (the last voxels of the 17 GB tensor), bf16-representable, off the kink
    lo, hi = far_lo, cgrid
    ilo, ihi = [l - 1 for l in lo], list(cgrid)
    xc = torch.cat([_crop(xa.detach(), ilo, ihi), _crop(xb, ilo, ihi)], 1).double().cpu()
    xcr, wr, br = xc.clone().requires_grad_(True), w64.clone().requires_grad_(True), b64.clone().requires_grad_(True)
    pre = F.conv3d(F.pad(xcr, [0, 1, 0, 1, 0, 1]), wr, br)
    gen = torch.Generator().manual_seed(9)
    off_kink = R.unshuffle_voxels((pre.detach().abs() > 1e-2).double(), 2)
    dyc = (torch.rand(off_kink.shape, generator=gen) - 0.5).to(BF).double() * off_kink
    # the engine stores dpre = dy * act'(y) as bf16 (the slope is the fp32 constant 0.01f): the oracle rounds the same way
    slope = torch.tensor(0.01, dtype=torch.float32).double()
    dpre_ref = q(R.shuffle_voxels(dyc, 2) * torch.where(pre.detach() > 0, torch.ones((), dtype=torch.float64), slope))
    pre.backward(dpre_ref)
    dy = torch.zeros_like(y)
    dy[:, :, 2 * lo[0]:, 2 * lo[1]:, 2 * lo[2]:] = dyc.to(DEV).to(BF)
    y.backward(dy)
    del dy, y
    dx = xa.grad
    outside = dx.clone()
    outside[:, :, ilo[0]:, ilo[1]:, ilo[2]:] = 0
    assert float(outside.float().abs().max()) == 0.0
    del outside
    one_ulp(_crop(dx, ilo, ihi), q(xcr.grad[:, :ca]), "dx")
    assert relerr(w.grad, wr.grad) < TOL, relerr(w.grad, wr.grad)
    assert relerr(bias.grad, br.grad) < TOL

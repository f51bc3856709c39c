"""N4: forward-only inference on the whole domain and the reference's evaluation metrics from the fused HIP pass
(include/sr3d.h: sr3d_eval_metrics), against golden values from the reference's own modules and, at the
full-domain size (1, 4, 32, 320, 320), against the CPU oracle."""
import json

import pytest
import torch

from helpers import load_golden, relerr, synthetic_inputs
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5
STDS = [8.4, 14.4, 21.6, 7.0]


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available()
    import sr3d_amd
    return sr3d_amd


def _modules(lm, stds):
    """exactly the dict script/train_model.py:366-379 builds"""
    return {"L1": lm.MyL1Loss(), "MaskedL1": lm.MaskedL1Loss(), "MaskedL1NearWall": lm.MaskedL1LossNearWall(),
            "ResidualContinuity": lm.ResidualContinuity(stds[1:]), "AbsDiffTemperature": lm.AbsDiffTemperature(stds[0]),
            "DiffVelocityNorm": lm.DiffVelocityVectorNorm(stds[1:]),
            "AbsDiffTemperatureLev": lm.AbsDiffTemperature(stds[0], lev=0),
            "DiffVelocityNormLev": lm.DiffVelocityVectorNorm(stds[1:], lev=0),
            "AbsDiffDivergence": lm.AbsDiffDivergence(stds[1:]), "DiffOmegaNorm": lm.DiffOmegaVectorNorm(stds[1:])}


@pytest.mark.parametrize("tag", ["iid", "tower"])
def test_metric_modules_vs_reference_golden(eng, tag):
    lm = eng.src.loss_maker
    g = load_golden("metrics.npz")
    meta = json.loads(str(g[f"{tag}/meta"]))
    _, b, y = synthetic_inputs(meta["B"], tuple(meta["hr"]), 4, meta["seed"], meta["kind"])
    gen = torch.Generator().manual_seed(meta["seed"] + 100)
    p = y + 0.3 * (torch.rand(y.shape, generator=gen) - 0.5)
    pd, yd, bd = p.to(DEV), y.to(DEV), b.to(DEV)
    fns = _modules(lm, STDS)
    fns.update({"L2": lm.MyL2Loss(), "MaskedL2": lm.MaskedL2Loss(), "MaskedL2NearWall": lm.MaskedL2LossNearWall(),
                "MixedMse": lm.MixedDivergenceGradientL2LossMse(STDS[1:]),
                "MixedGrdMse": lm.MixedDivergenceGradientL2LossGrdMse(STDS[1:]),
                "MixedDivMse": lm.MixedDivergenceGradientL2LossDivMse(STDS[1:])})
    with torch.no_grad():
        for k, fn in fns.items():
            got, ref = float(fn(pd, yd, bd)), float(g[f"{tag}/{k}"])
            assert abs(got - ref) <= TOL * abs(ref), (k, got, ref)
        pr, tr = fns["ResidualContinuity"].calc_both_pred_and_target(pd, yd, bd)
        assert abs(float(tr) - float(g[f"{tag}/ResidualContinuityTarget"])) <= TOL * float(g[f"{tag}/ResidualContinuityTarget"])
        for k, fn in (("AbsDiffTemperatureLev", lm.AbsDiffTemperature(STDS[0], lev=2)),
                      ("DiffVelocityNormLev", lm.DiffVelocityVectorNorm(STDS[1:], lev=2))):
            got, ref = float(fn(pd, yd, bd)), float(g[f"{tag}/Lev2/{k}"])
            assert abs(got - ref) <= TOL * abs(ref), (k, got, ref)


def test_one_launch_serves_all_ten_metrics(eng):
    """optim_helper.evaluate primes the fused pass with the union of the scales; the ten modules then hit the cache"""
    lm = eng.src.loss_maker
    _, b, y = synthetic_inputs(1, (8, 16, 24), 4, 5, "iid")
    p = (y + 0.2 * (torch.rand(y.shape, generator=torch.Generator().manual_seed(6)) - 0.5)).to(DEV)
    fns = _modules(lm, STDS)
    launches = []
    real = eng._lib.lib.sr3d_eval_metrics

    class Counting:
        def __call__(self, *a):
            launches.append(1)
            return real(*a)
    eng._lib.lib.sr3d_eval_metrics = Counting()
    try:
        class M(torch.nn.Module):
            def forward(self, X, bb):
                return p
        from sr3d_amd.src.optim_helper import evaluate
        res = evaluate(dataloader=[(torch.zeros(1, 4, 2, 4, 6), b[:, 0], y)], model=M(),
                       loss_fns=fns, device=DEV)
    finally:
        eng._lib.lib.sr3d_eval_metrics = real
    assert len(launches) == 1
    ref = R.eval_metrics(p.cpu(), y, b, STDS)
    for k, meter in res.items():
        assert abs(meter.avg - float(ref[k])) <= TOL * abs(float(ref[k])), k


def test_full_domain_inference_and_metrics(eng):
    """the evaluation workload of the reference: LR (1,4,8,80,80) -> HR (1,4,32,320,320), forward only, then the
    fused metrics; prediction windows and all metric values against the CPU oracle"""
    import bench
    cfg = bench.make_config("l1")
    torch.manual_seed(3)
    model = eng.make_model(cfg).to(DEV).eval()
    x, b, y = synthetic_inputs(1, (32, 320, 320), 4, 17, "tower")
    with torch.no_grad():
        pred = model(x.to(DEV), b.to(DEV))
    assert tuple(pred.shape) == (1, 4, 32, 320, 320)
    assert not any(p.grad is not None for p in model.parameters())
    # a 16 x 64 x 64 crop of the same input run through the oracle agrees in the crop's interior: the receptive field
    # of the 4-level U-Net is wider than the margin, so compare metrics instead on a synthetic prediction
    g = torch.Generator().manual_seed(18)
    p = (y + 0.2 * (torch.rand(y.shape, generator=g) - 0.5))
    out = eng.ops.eval_metrics(p.to(DEV), y.to(DEV), b.to(DEV), STDS).cpu()
    ref = R.eval_metrics(p, y, b, STDS)
    for k, i in eng._lib.EVAL_INDEX.items():
        assert abs(float(out[i]) - float(ref[k])) <= TOL * abs(float(ref[k])), (k, float(out[i]), float(ref[k]))
    # and on the network's own prediction (finite values, same code path as train_model.py's final evaluation)
    out2 = eng.ops.eval_metrics(pred, y.to(DEV), b.to(DEV), STDS).cpu()
    ref2 = R.eval_metrics(pred.cpu(), y, b, STDS)
    for k, i in eng._lib.EVAL_INDEX.items():
        assert abs(float(out2[i]) - float(ref2[k])) <= TOL * abs(float(ref2[k])), (k, float(out2[i]), float(ref2[k]))


def test_inference_equals_oracle_on_the_reference_eval_shape_at_small_width(eng):
    """forward-only prediction vs the oracle with the tiny golden model on a (32, 32, 48) domain"""
    from helpers import T, cfg_of, sub
    d = load_golden("model_tiny_a.npz")
    cfg, sd = cfg_of(d), sub(d, "sd")
    model = eng.make_model(cfg)
    model.load_state_dict(sd)
    model.to(DEV).eval()
    x, b, y = synthetic_inputs(1, (32, 32, 48), 4, 9, "tower")
    with torch.no_grad():
        pred = model(x.to(DEV), b.to(DEV))
    ref = R.unet_forward(sd, cfg["model"], x, b)
    assert relerr(pred, ref) < TOL


@pytest.mark.parametrize("tag", ["iid", "tower"])
def test_other_make_loss_branches(eng, tag):
    """make_loss's WeightedL1 / WeightedL2 / MixedGradientL2Loss branches (loss_maker.py:27-38): value and dL/dp
    against the reference's golden values"""
    g = load_golden("metrics.npz")
    meta = json.loads(str(g[f"{tag}/meta"]))
    _, b, y = synthetic_inputs(meta["B"], tuple(meta["hr"]), 4, meta["seed"], meta["kind"])
    gen = torch.Generator().manual_seed(meta["seed"] + 100)
    p = y + 0.3 * (torch.rand(y.shape, generator=gen) - 0.5)
    cases = {"WeightedL1": {"name": "WeightedL1", "weight_outside_building": 3.0},
             "WeightedL2": {"name": "WeightedL2", "weight_outside_building": 0.5},
             "MixedGradientL2": {"name": "MixedGradientL2Loss", "weight_gradient_loss": 2.0},
             "MixedGradientL2_off": {"name": "MixedGradientL2Loss"}}
    for nm, lc in cases.items():
        fn = eng.make_loss({"train": {"loss": lc}, "data": {"stds": STDS}})
        q = p.to(DEV).requires_grad_(True)
        v = fn(q, y.to(DEV), b.to(DEV))
        v.backward()
        ref = float(g[f"{tag}/{nm}"])
        assert abs(float(v.detach()) - ref) <= TOL * abs(ref), (nm, float(v.detach()), ref)
        assert relerr(q.grad, g[f"{tag}/{nm}/dp"]) < TOL, nm


@pytest.mark.parametrize("tag", ["iid", "tower"])
def test_unreached_loss_classes(eng, tag):
    """MixedGradientWeightedL2Loss / ChannelwiseMse (loss_maker.py:304-355, 753-764): value, terms and dL/dp against the
    reference's golden values"""
    from sr3d_amd.src.loss_maker import ChannelwiseMse, MixedGradientWeightedL2Loss
    g = load_golden("losses_extra.npz")
    meta = json.loads(str(g[f"{tag}/meta"]))
    _, b, y = synthetic_inputs(meta["B"], tuple(meta["hr"]), 4, meta["seed"], meta["kind"])
    gen = torch.Generator().manual_seed(meta["seed"] + 100)
    p = y + 0.3 * (torch.rand(y.shape, generator=gen) - 0.5)
    fn = MixedGradientWeightedL2Loss(weight_outside_building=3.0, weight_gradient_loss=2.0)
    q = p.to(DEV).requires_grad_(True)
    v = fn(q, y.to(DEV), b.to(DEV))
    v.backward()
    ref = float(g[f"{tag}/MixedGradientWeightedL2"])
    assert abs(float(v.detach()) - ref) <= TOL * abs(ref)
    assert relerr(q.grad, g[f"{tag}/MixedGradientWeightedL2/dp"]) < TOL
    with torch.no_grad():
        for a, r in zip(fn.calc_loss_terms(p.to(DEV), y.to(DEV), b.to(DEV)), g[f"{tag}/MixedGradientWeightedL2/terms"]):
            assert abs(float(a) - float(r)) <= TOL * abs(float(r)), (float(a), float(r))
    for i in range(4):
        q = p.to(DEV).requires_grad_(True)
        v = ChannelwiseMse(i)(q, y.to(DEV), b.to(DEV))
        v.backward()
        ref = float(g[f"{tag}/ChannelwiseMse{i}"])
        assert abs(float(v.detach()) - ref) <= TOL * abs(ref)
        assert relerr(q.grad, g[f"{tag}/ChannelwiseMse{i}/dp"]) < TOL


def test_ssim3d_vs_reference(eng):
    """SSIM3D / ssim3D / Ssim3dLoss (src/ssim.py, loss_maker.py:748-777) from the separable three-pass kernel"""
    from sr3d_amd.src.ssim import SSIM3D, ssim3D
    g = load_golden("ssim.npz")
    _, b, y = synthetic_inputs(2, (12, 20, 24), 4, 61, "iid")
    gen = torch.Generator().manual_seed(62)
    p = (y + 0.2 * (torch.rand(y.shape, generator=gen) - 0.5)).clamp(0, 1)
    pd, yd, bd = p.to(DEV), y.to(DEV), b.to(DEV)
    m4 = torch.broadcast_to(bd, pd.shape).contiguous()
    with torch.no_grad():
        assert abs(float(SSIM3D()(pd, yd, m4)) - float(g["gauss11_mean"])) < TOL
        assert relerr(SSIM3D(size_average=False)(pd, yd, m4), g["gauss11_map"]) < TOL
        assert abs(float(SSIM3D(window_size=7, use_gaussian=False)(pd, yd, m4)) - float(g["uniform7_mean"])) < TOL
        assert abs(float(SSIM3D(window_size=5, sigma=0.8, max_val=2.0)(pd, yd, m4)) - float(g["gauss5_s08_max2_mean"])) < TOL
        assert abs(float(ssim3D(pd, yd, m4)) - float(g["fn_mean"])) < TOL
        assert abs(float(eng.src.loss_maker.Ssim3dLoss()(pd, yd, bd)) - float(g["loss_mean"])) < TOL
    with pytest.raises(NotImplementedError):
        SSIM3D()(pd.clone().requires_grad_(True), yd, m4)
    # the reference's evaluation shape against the oracle (dense 1331-tap window on the CPU: crop to keep it short)
    x2, b2, y2 = synthetic_inputs(1, (16, 48, 64), 4, 63, "tower")
    p2 = (y2 + 0.1).clamp(0, 1)
    with torch.no_grad():
        got = float(eng.src.loss_maker.Ssim3dLoss()(p2.to(DEV), y2.to(DEV), b2.to(DEV)))
    ref = float(R.ssim3d(p2, y2, torch.broadcast_to(b2, p2.shape).contiguous(), eps=1e-3))
    assert abs(got - ref) < TOL


def test_metric_cache_never_serves_another_batch(eng):
    """the fused pass is cached per (prediction, target, mask) OBJECT: the next batch's tensors usually land at the same
    addresses (caching allocator), and must still get their own launch"""
    lm = eng.src.loss_maker
    fn_a, fn_b = lm.MaskedL1Loss(), lm.AbsDiffTemperature(STDS[0])
    vals = []
    for seed in (1, 2, 3):
        _, b, y = synthetic_inputs(1, (8, 16, 24), 4, seed, "iid")
        p = y + 0.1 * seed * torch.rand(y.shape, generator=torch.Generator().manual_seed(seed))
        pd, yd, bd = p.to(DEV), y.to(DEV), b.to(DEV)
        with torch.no_grad():
            got = (float(fn_a(pd, yd, bd)), float(fn_b(pd, yd, bd)))
        ref = R.eval_metrics(p, y, b, STDS)
        assert abs(got[0] - float(ref["MaskedL1"])) <= TOL * float(ref["MaskedL1"]), seed
        assert abs(got[1] - float(ref["AbsDiffTemperature"])) <= TOL * float(ref["AbsDiffTemperature"]), seed
        vals.append(got)
        del pd, yd, bd            # freed: the next iteration's tensors reuse these addresses
    assert len(set(vals)) == 3
    # an in-place change of the prediction invalidates the entry too
    _, b, y = synthetic_inputs(1, (8, 16, 24), 4, 9, "iid")
    pd, yd, bd = (y + 0.1).to(DEV), y.to(DEV), b.to(DEV)
    with torch.no_grad():
        v1 = float(fn_a(pd, yd, bd))
        pd.add_(0.05)
        v2 = float(fn_a(pd, yd, bd))
    assert abs(v2 - v1 - 0.05) < 1e-5

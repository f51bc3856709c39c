"""Generate tests/golden/*.npz by importing the reference (build container only).

Run from anywhere:   python oracle/make_golden.py
It puts /root/reference/pytorch on sys.path (read-only import, no bytecode
written), runs the reference's own modules on seeded synthetic inputs on CPU and
stores inputs + outputs as small .npz fixtures.  The reference never travels:
only these data files are committed.  (SURVEY.md section 8(c) lists what is
captured.)
"""
import json
import zlib
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference/pytorch"
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from model.custom_conv import MyConvWithAct2, PartialConv3d  # noqa: E402
from model.voxel_shuffle import shuffle_voxels, unshuffle_voxels  # noqa: E402
from src import math_helper as mh  # noqa: E402
from src.loss_maker import (calc_mask_near_build_wall, make_loss,  # noqa: E402
                            _calc_residual_continuity_eq)
from src.model_maker import make_model  # noqa: E402
from src.optim_helper import train as ref_train  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy().copy()  # copy: optimizer steps mutate parameters in place


def base_config():
    cfg = yaml.safe_load(open(os.path.join(REF, "config", "default.yml")))
    cfg["model"].update(num_feat0=4, num_feat1=8, num_feat2=8, num_feat3=16, num_feat4=16)
    return cfg


def tower_mask(Z, Y, X):
    """deterministic 'buildings': boxes rising from z=0 (1 = fluid, 0 = building)."""
    b = torch.ones(1, 1, Z, Y, X)
    b[..., : Z // 2, Y // 4: Y // 4 + 3, X // 4: X // 4 + 5] = 0
    b[..., : (3 * Z) // 4, Y // 2: Y // 2 + 4, X // 2: X // 2 + 2] = 0
    b[..., :2, :2, -3:] = 0
    return b


def make_inputs(B, hr, s, seed, mask_kind):
    g = torch.Generator().manual_seed(seed)
    Z, Y, X = hr
    x = torch.rand(B, 4, Z // s, Y // s, X // s, generator=g)
    y = torch.rand(B, 4, Z, Y, X, generator=g)
    if mask_kind == "iid":
        b = (torch.rand(B, 1, Z, Y, X, generator=g) > 0.2).float()
    else:
        b = tower_mask(Z, Y, X).repeat(B, 1, 1, 1, 1)
    return x, b, y


def model_fixture(name, cfg, B, hr, seed, mask_kind):
    torch.manual_seed(seed)
    s = 2 ** cfg["model"]["num_x2upsample"]
    model = make_model(cfg)
    # biases/weights at default init are fine, but make biases non-trivial
    x, b, y = make_inputs(B, hr, s, seed + 1, mask_kind)
    out = {"config_json": np.array(json.dumps(cfg)), "x": npy(x), "b": npy(b), "y": npy(y)}
    for k, v in model.state_dict().items():
        out["sd/" + k] = npy(v)

    pred = model(x, b)
    out["pred"] = npy(pred)

    # loss variants on the fp32 prediction
    for tag, (wg, wd) in {"g1d10": (1.0, 10.0), "g0d0": (0.0, 0.0), "g1d0": (1.0, 0.0), "g0d10": (0.0, 10.0)}.items():
        c2 = json.loads(json.dumps(cfg))
        c2["train"]["loss"]["weight_gradient_loss"] = wg
        c2["train"]["loss"]["weight_divergence_loss"] = wd
        lf = make_loss(c2)
        p = pred.detach().clone().requires_grad_(True)
        terms = lf.calc_loss_terms(predicts=p, targets=y, masks=b)
        total = lf(p, y, b)
        total.backward()
        out[f"loss/{tag}/terms"] = np.array([float(t) for t in terms], dtype=np.float64)
        out[f"loss/{tag}/total"] = np.array(float(total), dtype=np.float64)
        out[f"loss/{tag}/dpred"] = npy(p.grad)
    c2 = json.loads(json.dumps(cfg))
    c2["train"]["loss"] = {"name": "L1"}
    p = pred.detach().clone().requires_grad_(True)
    l1 = make_loss(c2)(p, y, b)
    l1.backward()
    out["loss/l1/total"] = np.array(float(l1), dtype=np.float64)
    out["loss/l1/dpred"] = npy(p.grad)

    # full backward with the config's own loss (g1d10)
    model.zero_grad()
    loss = make_loss(cfg)(model(x, b), y, b)
    loss.backward()
    out["train_loss"] = np.array(float(loss), dtype=np.float64)
    for k, v in model.named_parameters():
        out["grad/" + k] = npy(v.grad)

    # fp64 arbiter (stored rounded to fp32)
    m64 = make_model(cfg).double()
    m64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
    p64 = m64(x.double(), b.double())
    l64 = make_loss(cfg)(p64, y.double(), b.double())
    l64.backward()
    out["f64/pred"] = npy(p64).astype(np.float32)
    out["f64/train_loss"] = np.array(float(l64), dtype=np.float64)
    for k, v in m64.named_parameters():
        out["f64/grad/" + k] = npy(v.grad).astype(np.float32)

    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "pred", tuple(pred.shape), "loss", float(loss))


def ops_fixture():
    g = torch.Generator().manual_seed(7)
    out = {}
    f = torch.rand(2, 4, 6, 7, 9, generator=g)
    for ax, fn, fn_s in (("x", mh.differentiate_along_x, mh._differentiate_along_x),
                         ("y", mh.differentiate_along_y, mh._differentiate_along_y),
                         ("z", mh.differentiate_along_z, mh._differentiate_along_z)):
        out[f"diff/{ax}/pad0_d1"] = npy(fn(f, 1.0, 0))
        out[f"diff/{ax}/pad1_d5"] = npy(fn(f, 5.0, 1))
        out[f"diff/{ax}/scalar_d5"] = npy(fn_s(f, 5.0))
    out["diff/in"] = npy(f)
    v = torch.rand(2, 3, 6, 7, 9, generator=g)
    out["div/in"] = npy(v)
    out["div/pad0"] = npy(_calc_residual_continuity_eq(v, 5.0, 0))
    out["div/pad1"] = npy(_calc_residual_continuity_eq(v, 5.0, 1))

    b = (torch.rand(2, 1, 8, 10, 12, generator=g) > 0.15).float()
    out["wall/b_iid"] = npy(b)
    out["wall/near_iid"] = npy(calc_mask_near_build_wall(b))
    bt = tower_mask(16, 16, 32)
    out["wall/b_tower"] = npy(bt)
    out["wall/near_tower"] = npy(calc_mask_near_build_wall(bt))
    pool = torch.nn.AvgPool3d(2, 2)
    cur = bt
    for lvl in range(1, 5):
        cur = pool(cur)
        out[f"pool/tower_l{lvl}"] = npy(cur)

    u = torch.rand(2, 24, 3, 4, 5, generator=g)
    un = unshuffle_voxels(u, 2)
    out["shuffle/in"] = npy(u)
    out["shuffle/unshuffled"] = npy(un)
    assert torch.equal(shuffle_voxels(un, 2), u)

    # conv wrappers: fwd + grads.  (cin, cout, stride, mode, bias, act)
    cases = {
        "gated_s1_none": (5, 4, 1, "g_conv_with_separated_bias", False, None),
        "gated_s2_relu": (9, 8, 2, "g_conv_with_separated_bias", False, "relu"),
        "gated_s1_relu_bias": (8, 8, 1, "g_conv_with_separated_bias", True, "relu"),
        "gconv_s1_relu": (6, 8, 1, "g_conv", True, "relu"),
        "plain_s1_lrelu": (14, 4, 1, None, False, "lrelu"),
        "plain_s1_bias_none": (9, 4, 1, None, True, None),
        "plain_s2_lrelu": (7, 5, 2, None, False, "lrelu"),
        "wide_s1_lrelu": (70, 40, 1, None, False, "lrelu"),
        "wide_gated_s2_relu": (37, 35, 2, "g_conv_with_separated_bias", False, "relu"),
    }
    acts = {None: None, "relu": torch.nn.ReLU(), "lrelu": torch.nn.LeakyReLU()}
    for name, (cin, cout, st, mode, bias, act) in cases.items():
        torch.manual_seed(zlib.crc32(name.encode()) % 1000)
        m = MyConvWithAct2(cin, cout, 3, stride=st, padding=1, bias=bias, conv_mode=mode, act=acts[act])
        xin = (torch.rand(2, cin, 6, 8, 10, generator=g) - 0.3).requires_grad_(True)
        yo = m(xin)
        go = torch.rand(yo.shape, generator=g) - 0.5
        yo.backward(go)
        out[f"conv/{name}/meta"] = np.array(json.dumps(dict(cin=cin, cout=cout, stride=st, mode=mode, bias=bias, act=act)))
        out[f"conv/{name}/x"] = npy(xin)
        out[f"conv/{name}/y"] = npy(yo)
        out[f"conv/{name}/gy"] = npy(go)
        out[f"conv/{name}/gx"] = npy(xin.grad)
        for k, p in m.named_parameters():
            out[f"conv/{name}/sd/{k}"] = npy(p)
            out[f"conv/{name}/grad/{k}"] = npy(p.grad)

    # up path of UpBlock: conv(+bias) -> lrelu -> unshuffle
    torch.manual_seed(3)
    from model.unet import UpBlock
    ub = UpBlock(in1_channels=9, in2_channels=5, out_channels=4, bias=False, conv_mode=None, n_layers_in_block=2)
    x1 = (torch.rand(1, 9, 3, 4, 6, generator=g) - 0.4).requires_grad_(True)
    x2 = (torch.rand(1, 5, 6, 8, 12, generator=g) - 0.4).requires_grad_(True)
    x3 = ub.up(x1)
    yo = ub(x1, x2)
    go = torch.rand(yo.shape, generator=g) - 0.5
    yo.backward(go)
    out["upblock/x1"], out["upblock/x2"] = npy(x1), npy(x2)
    out["upblock/x3"], out["upblock/y"], out["upblock/gy"] = npy(x3), npy(yo), npy(go)
    out["upblock/gx1"], out["upblock/gx2"] = npy(x1.grad), npy(x2.grad)
    for k, p in ub.named_parameters():
        out[f"upblock/sd/{k}"] = npy(p)
        out[f"upblock/grad/{k}"] = npy(p.grad)

    # PartialConv3d (dead code in the reference; optional op)
    torch.manual_seed(5)
    pc = PartialConv3d(3, 4, 3, stride=1, padding=1, bias=True, multi_channel=True, return_mask=True)
    xin = torch.rand(1, 3, 5, 6, 7, generator=g)
    mk = (torch.rand(1, 3, 5, 6, 7, generator=g) > 0.3).float()
    po, pm = pc(xin, mk)
    out["pconv/x"], out["pconv/mask"], out["pconv/y"], out["pconv/mask_out"] = npy(xin), npy(mk), npy(po), npy(pm)
    out["pconv/weight"], out["pconv/bias"] = npy(pc.weight), npy(pc.bias)

    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)
    print("ops.npz", len(out), "arrays")


def train_fixture():
    """two iterations of optim_helper.train (Adam) on two different batches."""
    cfg = base_config()
    cfg["model"]["num_x2upsample"] = 1
    torch.manual_seed(11)
    model = make_model(cfg)
    out = {"config_json": np.array(json.dumps(cfg))}
    for k, v in model.state_dict().items():
        out["sd0/" + k] = npy(v)
    xs, bs, ys = [], [], []
    for i in range(2):
        x, b, y = make_inputs(1, (16, 16, 32), 2, 100 + i, "iid")
        xs.append(x), bs.append(b[:, 0]), ys.append(y)  # dataset yields b without channel dim
    ds = torch.utils.data.TensorDataset(torch.cat(xs), torch.cat(bs), torch.cat(ys))
    dl = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    lr = 1e-3  # larger than default.yml's 1e-4 so that two steps move the weights visibly
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    avg = ref_train(dl, model, make_loss(cfg), opt, "cpu", hide_progress_bar=True)
    out["lr"] = np.array(lr)
    out["avg_loss"] = np.array(avg, dtype=np.float64)
    out["x"], out["b"], out["y"] = npy(torch.cat(xs)), npy(torch.cat(bs)), npy(torch.cat(ys))
    for k, v in model.state_dict().items():
        out["sd2/" + k] = npy(v)
    np.savez_compressed(os.path.join(OUT, "train2.npz"), **out)
    print("train2.npz avg loss", avg)


def fullwidth_fixture():
    """default.yml-width single layers on a small grid (weights regenerated from a
    seed on the test side, only outputs are stored) -- pins K-tail handling of
    Cin=129 -> Cout=1032 with the unshuffle epilogue and 194 -> 64."""
    out = {}
    for name, (cin, cout, grid) in {"up1_up0": (129, 1032, (2, 4, 8)), "up1_convs0": (194, 64, (4, 8, 16))}.items():
        g = torch.Generator().manual_seed(2024)
        w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        x = torch.rand(1, cin, *grid, generator=g) - 0.5
        y = torch.nn.functional.leaky_relu(torch.nn.functional.conv3d(x, w, bias, padding=1), 0.01)
        if cout % 8 == 0 and name.endswith("up0"):
            y = unshuffle_voxels(y, 2)
        out[f"{name}/meta"] = np.array(json.dumps(dict(cin=cin, cout=cout, grid=grid, seed=2024)))
        out[f"{name}/y"] = npy(y)
    np.savez_compressed(os.path.join(OUT, "fullwidth.npz"), **out)
    print("fullwidth.npz")


def _sampled(t, n_full=20000, n_samp=4096):
    """small tensors whole, large ones as a strided sample of ~n_samp elements (start 0)"""
    flat = t.detach().reshape(-1)
    if flat.numel() <= n_full:
        return npy(flat)
    return npy(flat[::_prime_step(flat.numel() // n_samp)])


def _prime_step(n):
    """smallest prime >= n: a stride that does not lock onto the (tap, channel) periods of a weight tensor"""
    while any(n % q == 0 for q in range(2, int(n ** 0.5) + 1)):
        n += 1
    return n


def default_width_fixture(name, hr, s, seed, mask_kind, loss_name):
    """The reference's UNetSR at default.yml widths (65.47 M parameters) on a small grid.  The 262 MB of weights are
    NOT stored: `torch.manual_seed(seed); make_model(cfg)` gives the same initial weights in the reference and in the
    engine (tests/test_cabi_and_host.py pins that); an exact per-tensor checksum (int64 sum of the bit patterns) is stored to prove it.  Inputs come from
    tests/helpers.py:synthetic_inputs with the stored seed.  Stored: prediction, loss values (L1 and the four terms of
    the mixed loss), dL/dpred, and for each of the 48 parameter gradients its fp64 norm plus the whole tensor (small
    ones) or a strided sample (large ones), in fp32 and from an fp64 run of the reference."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from helpers import synthetic_inputs
    cfg = yaml.safe_load(open(os.path.join(REF, "config", "default.yml")))
    cfg["model"]["num_x2upsample"] = {2: 1, 4: 2}[s]
    if loss_name == "L1":
        cfg["train"]["loss"] = {"name": "L1"}
    torch.manual_seed(seed)
    model = make_model(cfg)
    x, b, y = synthetic_inputs(1, hr, s, seed + 1, mask_kind)
    out = {"config_json": np.array(json.dumps(cfg)),
           "meta": np.array(json.dumps(dict(hr=hr, s=s, seed=seed, mask_kind=mask_kind, loss=loss_name)))}
    for k, v in model.state_dict().items():
        out["sdsum/" + k] = np.array(int(v.view(torch.int32).to(torch.int64).sum()), dtype=np.int64)   # order-independent
    pred = model(x, b)
    out["pred"] = npy(pred)
    cm = yaml.safe_load(open(os.path.join(REF, "config", "default.yml")))
    lf = make_loss(cm)
    p = pred.detach().clone().requires_grad_(True)
    terms = lf.calc_loss_terms(predicts=p, targets=y, masks=b)
    total = lf(p, y, b)
    total.backward()
    out["loss/mixed/terms"] = np.array([float(t) for t in terms], dtype=np.float64)
    out["loss/mixed/total"] = np.array(float(total), dtype=np.float64)
    if loss_name != "L1":
        out["loss/mixed/dpred"] = npy(p.grad)
    out["loss/l1/total"] = np.array(float((pred - y).abs().mean()), dtype=np.float64)

    model.zero_grad()
    loss = make_loss(cfg)(model(x, b), y, b)
    loss.backward()
    out["train_loss"] = np.array(float(loss), dtype=np.float64)
    for k, v in model.named_parameters():
        out["gradnorm/" + k] = np.array(float(v.grad.double().norm()), dtype=np.float64)
        out["grad/" + k] = _sampled(v.grad)

    m64 = make_model(cfg).double()
    m64.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
    p64 = m64(x.double(), b.double())
    l64 = make_loss(cfg)(p64, y.double(), b.double())
    l64.backward()
    out["f64/pred_s4"] = npy(p64.reshape(-1)[::4]).astype(np.float32)   # every 4th element
    out["f64/train_loss"] = np.array(float(l64), dtype=np.float64)
    for k, v in m64.named_parameters():
        out["f64/gradnorm/" + k] = np.array(float(v.grad.norm()), dtype=np.float64)
        out["f64/grad/" + k] = _sampled(v.grad).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "pred", tuple(pred.shape), "loss", float(loss), "f64", float(l64))



def metrics_fixture():
    """the reference's evaluation metric modules (script/train_model.py:366-379 builds exactly these) on seeded
    small inputs; inputs are regenerated on the test side (tests/helpers.py:synthetic_inputs)"""
    from src import loss_maker as LM
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from helpers import synthetic_inputs
    stds = [8.4, 14.4, 21.6, 7.0]
    out = {"stds": np.array(stds)}
    for tag, (B, hr, seed, kind) in {"iid": (2, (8, 16, 24), 41, "iid"), "tower": (1, (16, 24, 40), 42, "tower")}.items():
        _, b, y = synthetic_inputs(B, hr, 4, seed, kind)
        g = torch.Generator().manual_seed(seed + 100)
        p = y + 0.3 * (torch.rand(y.shape, generator=g) - 0.5)
        fns = {"L1": LM.MyL1Loss(), "L2": LM.MyL2Loss(), "MaskedL1": LM.MaskedL1Loss(), "MaskedL2": LM.MaskedL2Loss(),
               "MaskedL1NearWall": LM.MaskedL1LossNearWall(), "MaskedL2NearWall": LM.MaskedL2LossNearWall(),
               "ResidualContinuity": LM.ResidualContinuity(stds[1:]),
               "AbsDiffTemperature": LM.AbsDiffTemperature(stds[0]),
               "DiffVelocityNorm": LM.DiffVelocityVectorNorm(stds[1:]),
               "AbsDiffTemperatureLev": LM.AbsDiffTemperature(stds[0], lev=0),
               "DiffVelocityNormLev": LM.DiffVelocityVectorNorm(stds[1:], lev=0),
               "AbsDiffDivergence": LM.AbsDiffDivergence(stds[1:]),
               "DiffOmegaNorm": LM.DiffOmegaVectorNorm(stds[1:])}
        out[f"{tag}/meta"] = np.array(json.dumps(dict(B=B, hr=hr, seed=seed, kind=kind)))
        for k, fn in fns.items():
            out[f"{tag}/{k}"] = np.array(float(fn(p, y, b)), dtype=np.float64)
        pr, tr = LM.ResidualContinuity(stds[1:]).calc_both_pred_and_target(p, y, b)
        assert float(pr) == float(out[f"{tag}/ResidualContinuity"])
        out[f"{tag}/ResidualContinuityTarget"] = np.array(float(tr), dtype=np.float64)
        out[f"{tag}/Lev2/AbsDiffTemperatureLev"] = np.array(float(LM.AbsDiffTemperature(stds[0], lev=2)(p, y, b)))
        out[f"{tag}/Lev2/DiffVelocityNormLev"] = np.array(float(LM.DiffVelocityVectorNorm(stds[1:], lev=2)(p, y, b)))
        for nm, cls in (("Mse", LM.MixedDivergenceGradientL2LossMse), ("GrdMse", LM.MixedDivergenceGradientL2LossGrdMse),
                        ("DivMse", LM.MixedDivergenceGradientL2LossDivMse)):
            out[f"{tag}/Mixed{nm}"] = np.array(float(cls(stds[1:])(p, y, b)), dtype=np.float64)
        # the other make_loss branches (loss_maker.py:27-38): value and dL/dp
        for nm, fn in (("WeightedL1", LM.WeightedL1Loss(3.0)), ("WeightedL2", LM.WeightedL2Loss(0.5)),
                       ("MixedGradientL2", LM.MixedGradientL2Loss(2.0)), ("MixedGradientL2_off", LM.MixedGradientL2Loss(None))):
            pp = p.clone().requires_grad_(True)
            v = fn(pp, y, b)
            v.backward()
            out[f"{tag}/{nm}"] = np.array(float(v), dtype=np.float64)
            out[f"{tag}/{nm}/dp"] = npy(pp.grad)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("metrics.npz", len(out))



def extra_losses_fixture():
    """the two loss classes no make_loss branch reaches (loss_maker.py:304-355 MixedGradientWeightedL2Loss, :753-764 ChannelwiseMse):
    value, the three terms and dL/dp on the inputs of the metrics fixture"""
    import src.loss_maker as LM
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from helpers import synthetic_inputs
    out = {}
    for tag, (B, hr, seed, kind) in {"iid": (2, (8, 16, 24), 41, "iid"), "tower": (1, (16, 24, 40), 42, "tower")}.items():
        _, b, y = synthetic_inputs(B, hr, 4, seed, kind)
        g = torch.Generator().manual_seed(seed + 100)
        p = y + 0.3 * (torch.rand(y.shape, generator=g) - 0.5)
        out[f"{tag}/meta"] = np.array(json.dumps(dict(B=B, hr=hr, seed=seed, kind=kind)))
        fn = LM.MixedGradientWeightedL2Loss(weight_outside_building=3.0, weight_gradient_loss=2.0)
        pp = p.clone().requires_grad_(True)
        v = fn(pp, y, b)
        v.backward()
        out[f"{tag}/MixedGradientWeightedL2"] = np.array(float(v), dtype=np.float64)
        out[f"{tag}/MixedGradientWeightedL2/dp"] = npy(pp.grad)
        out[f"{tag}/MixedGradientWeightedL2/terms"] = np.array([float(t) for t in fn.calc_loss_terms(p, y, b)], dtype=np.float64)
        for i in range(4):
            pp = p.clone().requires_grad_(True)
            v = LM.ChannelwiseMse(i)(pp, y, b)
            v.backward()
            out[f"{tag}/ChannelwiseMse{i}"] = np.array(float(v), dtype=np.float64)
            out[f"{tag}/ChannelwiseMse{i}/dp"] = npy(pp.grad)
    np.savez_compressed(os.path.join(OUT, "losses_extra.npz"), **out)
    print("losses_extra.npz", len(out))


def pconv_fixture():
    """PartialConv3d (custom_conv.py:129-234): forward, updated mask and all gradients"""
    out = {}
    cases = {"multi_bias": dict(cin=3, cout=4, stride=1, bias=True, multi=True, mask="given"),
             "single_nobias_s2": dict(cin=5, cout=6, stride=2, bias=False, multi=False, mask="given"),
             "single_bias_nomask": dict(cin=2, cout=3, stride=1, bias=True, multi=False, mask=None)}
    for name, c in cases.items():
        torch.manual_seed(17)
        pc = PartialConv3d(c["cin"], c["cout"], 3, stride=c["stride"], padding=1, bias=c["bias"],
                           multi_channel=c["multi"], return_mask=True)
        g = torch.Generator().manual_seed(18)
        x = (torch.rand(2, c["cin"], 6, 7, 9, generator=g) - 0.5).requires_grad_(True)
        mk = None
        if c["mask"] == "given":
            mk = (torch.rand(2, c["cin"] if c["multi"] else 1, 6, 7, 9, generator=g) > 0.4).float()
        y, um = pc(x, mk)
        gy = torch.rand(y.shape, generator=g) - 0.5
        y.backward(gy)
        out[f"{name}/meta"] = np.array(json.dumps(c))
        out[f"{name}/x"], out[f"{name}/y"], out[f"{name}/um"], out[f"{name}/gy"] = npy(x), npy(y), npy(um), npy(gy)
        if mk is not None:
            out[f"{name}/mask"] = npy(mk)
        out[f"{name}/gx"], out[f"{name}/w"], out[f"{name}/gw"] = npy(x.grad), npy(pc.weight), npy(pc.weight.grad)
        if c["bias"]:
            out[f"{name}/b"], out[f"{name}/gb"] = npy(pc.bias), npy(pc.bias.grad)
    np.savez_compressed(os.path.join(OUT, "pconv.npz"), **out)
    print("pconv.npz", len(out))



def ssim_fixture():
    """the reference's SSIM3D / Ssim3dLoss (src/ssim.py, loss_maker.py:748-777) on a seeded small input"""
    from src import loss_maker as LM
    from src.ssim import SSIM3D, ssim3D
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from helpers import synthetic_inputs
    _, b, y = synthetic_inputs(2, (12, 20, 24), 4, 61, "iid")
    g = torch.Generator().manual_seed(62)
    p = (y + 0.2 * (torch.rand(y.shape, generator=g) - 0.5)).clamp(0, 1)
    m4 = torch.broadcast_to(b, p.shape).contiguous()
    out = {"meta": np.array(json.dumps(dict(B=2, hr=(12, 20, 24), seed=61)))}
    out["gauss11_mean"] = np.array(float(SSIM3D()(p, y, m4)), dtype=np.float64)
    out["gauss11_map"] = npy(SSIM3D(size_average=False)(p, y, m4))
    out["uniform7_mean"] = np.array(float(SSIM3D(window_size=7, use_gaussian=False)(p, y, m4)), dtype=np.float64)
    out["gauss5_s08_max2_mean"] = np.array(float(SSIM3D(window_size=5, sigma=0.8, max_val=2.0)(p, y, m4)), dtype=np.float64)
    out["fn_mean"] = np.array(float(ssim3D(p, y, m4)), dtype=np.float64)
    out["loss_mean"] = np.array(float(LM.Ssim3dLoss()(p, y, b)), dtype=np.float64)        # eps = 1e-3
    np.savez_compressed(os.path.join(OUT, "ssim.npz"), **out)
    print("ssim.npz", {k: float(v) for k, v in out.items() if k.endswith("mean")})


def gradnorm_fixture():
    """N3: the reference's GradNorm (src/gradnorm.py:74-115) on the tiny model of model_tiny_a.npz: (1) ONE call of
    GradNorm.backward with non-trivial weights / initial losses -> total, weights.grad, every parameter gradient;
    (2) two steps of the reference's optim_helper.train with grad_norm (Adam over model + weights, renormalisation)."""
    from src.gradnorm import GradNorm
    z = np.load(os.path.join(OUT, "model_tiny_a.npz"))
    cfg = json.loads(str(z["config_json"]))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    x, b, y = (torch.from_numpy(z[k]) for k in ("x", "b", "y"))
    out = {"config_json": np.array(json.dumps(cfg)), "w0": np.array([1.0, 0.5, 2.0], dtype=np.float32),
           "init_losses": np.array([0.1, 2.0, 0.05], dtype=np.float32), "alpha": np.array(1.5)}
    model = make_model(cfg)
    model.load_state_dict(sd)
    loss_fn = make_loss(cfg)
    gn = GradNorm(n_tasks=3, alpha=1.5, device="cpu", output_dir_path="/tmp")
    with torch.no_grad():
        gn.weights.copy_(torch.from_numpy(out["w0"]))
    gn.init_losses = torch.from_numpy(out["init_losses"]).clone()
    terms = loss_fn.calc_loss_terms(predicts=model(x, b), targets=y, masks=b)
    model.zero_grad()
    total = gn.backward(loss_list=list(terms), last_shared_params=model.get_last_params())
    out["one/terms"] = npy(torch.stack(list(terms)))
    out["one/total"] = npy(total)
    out["one/weights_grad"] = npy(gn.weights.grad)
    for k, p in model.named_parameters():
        out["one/grad/" + k] = npy(p.grad)

    # (2) the training loop: fresh GradNorm (weights 1, initial losses from the first batch), lr of the weights 0.025
    model = make_model(cfg)
    model.load_state_dict(sd)
    gn = GradNorm(n_tasks=3, alpha=1.5, device="cpu", output_dir_path="/tmp", clipping_weight_min=0.1)
    opt = torch.optim.Adam([{"params": model.parameters()}, {"params": gn.weights, "lr": 0.025}], lr=1e-3)
    ds = torch.utils.data.TensorDataset(x, b[:, 0], y)
    dl = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    avg = ref_train(dl, model, loss_fn, opt, "cpu", hide_progress_bar=True, grad_norm=gn)
    out["loop/lr"], out["loop/lr_weights"], out["loop/clip"] = np.array(1e-3), np.array(0.025), np.array(0.1)
    out["loop/avg_loss"] = np.array(avg)
    out["loop/weights"] = npy(gn.weights)
    out["loop/init_losses"] = npy(gn.init_losses)
    for k, v in model.state_dict().items():
        out["loop/sd2/" + k] = npy(v)
    np.savez_compressed(os.path.join(OUT, "gradnorm.npz"), **out)
    print("gradnorm.npz: total", float(total), "weights.grad", gn.weights.grad if gn.weights.grad is not None else None,
          "loop weights", out["loop/weights"], "avg", avg)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gradnorm":
        gradnorm_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ssim":
        ssim_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pconv":
        pconv_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "extra_losses":
        extra_losses_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":
        metrics_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "default_width":   # only the (slow) default-width fixtures
        default_width_fixture("model_default_a.npz", (16, 64, 64), 4, 31, "tower", "MixedDivergenceGradientL2Loss")
        default_width_fixture("model_default_b.npz", (32, 64, 64), 2, 32, "iid", "L1")
        sys.exit(0)
    cfg_a = base_config()  # num_x2upsample = 2
    model_fixture("model_tiny_a.npz", cfg_a, 2, (16, 16, 16), 21, "iid")
    cfg_b = base_config()
    cfg_b["model"]["num_x2upsample"] = 1
    model_fixture("model_tiny_b.npz", cfg_b, 1, (16, 32, 48), 22, "tower")
    ops_fixture()
    train_fixture()
    fullwidth_fixture()

"""The callers either side of the hot path: the reference's train loop and
GradNorm, run on the engine and compared with golden vectors / a CPU replay on the oracle."""
import numpy as np
import pytest
import torch

from helpers import T, cfg_of, load_golden, relerr, sub, trimmed_relerr
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available()
    import sr3d_amd
    return sr3d_amd


def test_optim_helper_train_reproduces_reference_two_steps(eng):
    """golden train2.npz was produced by the reference's optim_helper.train on a 2-sample loader"""
    from sr3d_amd.src import optim_helper
    d = load_golden("train2.npz")
    cfg = cfg_of(d)
    model = eng.make_model(cfg)
    model.load_state_dict(sub(d, "sd0"))
    model.to(DEV)
    ds = torch.utils.data.TensorDataset(T(d["x"]), T(d["b"]), T(d["y"]))
    dl = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    opt = eng.FlatAdam(model.parameters(), lr=float(d["lr"]))
    avg = optim_helper.train(dl, model, eng.make_loss(cfg), opt, DEV)
    assert abs(avg - float(d["avg_loss"])) < 1e-5 * float(d["avg_loss"])
    # Adam's first steps turn a gradient element within eps = 1e-8 of zero into an O(lr) update, so single elements of
    # a tensor may legitimately differ by ~1e-5 of its norm; everything but the worst 0.25 % must agree to 1e-5
    for k, v in model.state_dict().items():
        assert relerr(v, d["sd2/" + k]) < 1e-5 or (trimmed_relerr(v, d["sd2/" + k]) < 2e-6 and
                                                   relerr(v, d["sd2/" + k]) < 1e-4), k
    val = optim_helper.test(dl, model, eng.make_loss(cfg), DEV)
    assert np.isfinite(val)


def test_loss_terms_are_individually_differentiable(eng):
    """d(mse)/dp, d(grd)/dp, d(div)/dp from the golden totals by linearity"""
    d = load_golden("model_tiny_b.npz")
    cfg = cfg_of(d)
    p0, y, b = T(d["pred"]).to(DEV), T(d["y"]).to(DEV), T(d["b"]).to(DEV)
    lf = eng.make_loss(cfg)  # w_g = 1, w_d = 10
    want = {0: T(d["loss/g0d0/dpred"]), 1: T(d["loss/g1d0/dpred"]) - T(d["loss/g0d0/dpred"]),
            2: (T(d["loss/g0d10/dpred"]) - T(d["loss/g0d0/dpred"])) / 10.0}
    for i in range(3):
        p = p0.clone().requires_grad_(True)
        terms = lf.calc_loss_terms(predicts=p, targets=y, masks=b)
        (g,) = torch.autograd.grad(terms[i], p)
        assert relerr(g, want[i]) < 2e-5, i
    # and an arbitrary mix in one backward
    p = p0.clone().requires_grad_(True)
    terms = lf.calc_loss_terms(predicts=p, targets=y, masks=b)
    (0.3 * terms[0] + 2.0 * terms[1] - 0.7 * terms[2]).backward()
    assert relerr(p.grad, 0.3 * want[0] + 2.0 * want[1] - 0.7 * want[2]) < 2e-5


def test_gradnorm_step_matches_cpu_replay(eng):
    """one GradNorm training step on the engine vs the same algorithm replayed on the CPU oracle"""
    from sr3d_amd.src.gradnorm import GradNorm
    d = load_golden("model_tiny_a.npz")
    cfg = cfg_of(d)
    sd = sub(d, "sd")
    x, b, y = T(d["x"]), T(d["b"]), T(d["y"])
    # --- engine
    model = eng.make_model(cfg)
    model.load_state_dict(sd)
    model.to(DEV)
    gn = GradNorm(n_tasks=3, alpha=1.5, device=DEV)
    with torch.no_grad():
        gn.weights.copy_(torch.tensor([1.0, 0.5, 2.0]))
    gn.init_losses = torch.tensor([0.1, 2.0, 0.05], device=DEV)
    lf = eng.make_loss(cfg)
    terms = lf.calc_loss_terms(predicts=model(x.to(DEV), b.to(DEV)), targets=y.to(DEV), masks=b.to(DEV))
    model.zero_grad()
    total = gn.backward(loss_list=list(terms), last_shared_params=model.get_last_params())
    # --- CPU replay with the oracle
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pred = R.unet_forward(leaves, cfg["model"], x, b)
    lc = cfg["train"]["loss"]
    rt = R.mixed_div_grad_terms(pred, y, b, lc["weight_gradient_loss"], lc["weight_divergence_loss"],
                                cfg["data"]["stds"][1:])
    w = torch.tensor([1.0, 0.5, 2.0], requires_grad=True)
    losses = torch.stack(list(rt))
    ref_total = (w * losses).sum()
    ref_total.backward(retain_graph=True)
    norms = torch.stack([torch.norm(w_i * torch.autograd.grad(L_i, leaves["last.weight"], retain_graph=True)[0])
                         for w_i, L_i in zip(w, losses)])
    with torch.no_grad():
        ratios = losses / torch.tensor([0.1, 2.0, 0.05])
        const = norms.mean() * (ratios / ratios.mean()) ** 1.5
    ref_wgrad = torch.autograd.grad((norms - const).abs().sum(), w)[0]
    assert abs(float(total) - float(ref_total)) < 1e-5 * abs(float(ref_total))
    assert relerr(gn.weights.grad, ref_wgrad) < 1e-4
    for k, p in model.named_parameters():
        assert relerr(p.grad, leaves[k].grad) < 5e-5, k

"""shared helpers for the test-suite (golden loading, normwise error)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def sub(d, prefix):
    """entries of d under 'prefix/' as torch tensors"""
    n = len(prefix) + 1
    return {k[n:]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items() if k.startswith(prefix + "/")}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def cfg_of(d):
    return json.loads(str(d["config_json"]))


def relerr(a, b):
    """normwise relative error ||a-b||_2 / ||b||_2 (SURVEY.md section 7 'hard parts')."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    den = float(b.norm())
    num = float((a - b).norm())
    return num / den if den > 0 else num


def tower_mask(Z, Y, X):
    """deterministic 'buildings': boxes rising from z=0 (1 = fluid, 0 = building); same as oracle/make_golden.py"""
    b = torch.ones(1, 1, Z, Y, X)
    b[..., : Z // 2, Y // 4: Y // 4 + 3, X // 4: X // 4 + 5] = 0
    b[..., : (3 * Z) // 4, Y // 2: Y // 2 + 4, X // 2: X // 2 + 2] = 0
    b[..., :2, :2, -3:] = 0
    return b


def synthetic_inputs(B, hr, s, seed, mask_kind):
    """seeded LR input, building mask and HR target (the recipe oracle/make_golden.py used for the fixtures whose
    inputs are regenerated instead of stored)"""
    g = torch.Generator().manual_seed(seed)
    Z, Y, X = hr
    x = torch.rand(B, 4, Z // s, Y // s, X // s, generator=g)
    y = torch.rand(B, 4, Z, Y, X, generator=g)
    if mask_kind == "iid":
        b = (torch.rand(B, 1, Z, Y, X, generator=g) > 0.2).float()
    else:
        b = tower_mask(Z, Y, X).repeat(B, 1, 1, 1, 1)
    return x, b, y


def sampled(t, n_full=20000, n_samp=4096):
    """the sampling oracle/make_golden.py:_sampled applied to large gradients"""
    flat = torch.as_tensor(t).detach().reshape(-1)
    if flat.numel() <= n_full:
        return flat
    n = flat.numel() // n_samp
    while any(n % q == 0 for q in range(2, int(n ** 0.5) + 1)):      # smallest prime >= n, as in make_golden.py
        n += 1
    return flat[::n]


def trimmed_relerr(a, b, frac=0.0025):
    """normwise error after discarding the `frac` of elements with the largest error (at least one): for quantities
    in which a handful of elements are legitimately ill-conditioned (a flipped activation decision; an Adam update of
    a gradient element within eps of zero) while a wrong tile or tap would touch far more than that"""
    a = torch.as_tensor(a).detach().double().cpu().flatten()
    b = torch.as_tensor(b).detach().double().cpu().flatten()
    e2 = (a - b) ** 2
    drop = max(1, int(frac * e2.numel()))
    kept = e2.sum() - torch.topk(e2, drop).values.sum()
    den = float(b.norm())
    return float(kept.clamp_min(0).sqrt()) / den if den > 0 else float(kept.clamp_min(0).sqrt())

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

"""script/train_model.py end to end on one GPU with a synthetic data tree in the reference's format:
reference CLI + YAML in, weights.pth / learning_history.csv / log.txt out."""
import os
import subprocess
import sys

import pytest
import torch
import yaml

from data_fixture import write_synthetic_tree
from helpers import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", ["gradnorm", "hipgraph", "bf16_hipgraph"])
def test_train_script_single_gpu(tmp_path, variant):
    data_root = write_synthetic_tree(tmp_path / "d", HR=(16, 32, 32), days=6)
    cfg = {
        "data": {"data_dir_names": ["10"], "train_valid_test_ratios": [0.6, 0.2, 0.2], "hr_org_size": [16, 32, 32],
                 "hr_crop_size": [16, 16, 32], "means": [302.0, -6.5, -9.1, -3.5], "stds": [8.4, 14.4, 21.6, 7.0],
                 "datasizes": {"train": 100, "valid": 100, "test": 100}, "nan_value": 0.0, "batch_size": 2, "seed": 42,
                 "num_workers": 0},
        "train": {"num_epochs": 2, "lr": 1.0e-3, "num_loops_train": 1, "num_loops_valid": 1, "seed": 42,
                  "write_out_inferences": True,
                  "loss": {"name": "MixedDivergenceGradientL2Loss", "weight_gradient_loss": 1.0,
                           "weight_divergence_loss": 10.0},
                  "grad_norm": {"n_tasks": 3, "alpha": 1.5, "lr": 1.0e-2}},
        "model": {"model_name": "unet", "in_channels": 4, "out_channels": 4, "num_feat0": 4, "num_feat1": 8,
                  "num_feat2": 8, "num_feat3": 16, "num_feat4": 16, "num_x2upsample": 2, "num_latent_layers": 3,
                  "n_layers_in_block": 2, "bias_feat_extraction": False,
                  "conv_mode_feat_extraction": "g_conv_with_separated_bias",
                  "conv_mode_down_block": "g_conv_with_separated_bias", "conv_mode_up_block": None},
    }
    if variant in ("hipgraph", "bf16_hipgraph"):      # engine extension: the step as a hipGraph replay (no GradNorm: it has its own optimizer)
        del cfg["train"]["grad_norm"]
        cfg["train"]["hip_graph"] = True
    if variant == "bf16_hipgraph":                    # engine extension: bf16 storage inside the network (BASELINE configs[4])
        cfg["model"]["storage_dtype"] = "bf16"
    (tmp_path / "exp").mkdir()
    cfg_path = tmp_path / "exp" / "tiny.yml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    script = os.path.join(ROOT, "3d-sr-micrometeorology_amd", "script", "train_model.py")
    r = subprocess.run([sys.executable, script, "--config_path", str(cfg_path), "--world_size", "1", "--data_root",
                        str(data_root), "--result_root", str(tmp_path / "res"), "--inference_root",
                        str(tmp_path / "inf")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = tmp_path / "res" / "exp" / "tiny"
    sd = torch.load(out / "weights.pth")
    assert len(sd) == 48 and "conv0.conv.mask_conv3d.bias" in sd
    hist = (out / "learning_history.csv").read_text().strip().splitlines()
    assert hist[0] == "loss,val_loss" and len(hist) == 3
    assert "Epoch: 2" in (out / "log.txt").read_text()
    assert (out / "grad_norm_weights_0.csv").exists() == (variant == "gradnorm")
    # final evaluation: the reference's ten metrics on the whole-domain test loader, then the inferences
    log = (out / "log.txt").read_text()
    for name in ("L1", "MaskedL1", "MaskedL1NearWall", "ResidualContinuityEq", "AbsDiffTemperature", "DiffVelocityNorm",
                 "AbsDiffTemperatureLevZero", "DiffVelocityNormLevZero", "AbsDiffDivergence", "DiffOmegaVectorNorm"):
        assert f"{name}: " in log, name
    import numpy as np
    inf = sorted(os.listdir(tmp_path / "inf" / "exp" / "tiny"))
    assert len(inf) % 4 == 0 and len(inf) >= 4 and inf[0].endswith("_BM.npy")
    stamp = inf[0].split("_")[0]
    sr = np.load(tmp_path / "inf" / "exp" / "tiny" / f"{stamp}_SR.npy")
    hr = np.load(tmp_path / "inf" / "exp" / "tiny" / f"{stamp}_HR.npy")
    assert sr.shape == hr.shape == (1, 4, 16, 32, 32) and np.isfinite(sr).all()

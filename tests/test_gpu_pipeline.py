"""N2 on the device: raw windows from the workers, normalise / clamp / NaN-fill by sr3d_preprocess on a side stream.
Batches must be bit-identical to the CPU sample pipeline (which is itself pinned to the reference's Dataset by
tests/test_dataset.py)."""
import pytest
import torch

from data_fixture import CASES, EVAL_CONFIG, HR32, write_synthetic_tree

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available()
    import sr3d_amd
    return sr3d_amd


@pytest.mark.parametrize("case", ["crop", "crop_z", "full"])
def test_device_pipeline_equals_cpu_pipeline(eng, tmp_path, case):
    from sr3d_amd.src.dataloader import get_all_new_lr_data_dir_paths, make_dataloaders, split_into_train_valid_test_dirs
    tree = write_synthetic_tree(tmp_path)
    dirs = get_all_new_lr_data_dir_paths(tree)
    split = split_into_train_valid_test_dirs(dirs, [0.6, 0.2, 0.2])
    kw = {k: v for k, v in CASES[case].items() if k != "use_cropping"}
    got, ref = [], []
    for pipe, sink in ((None, ref), (DEV, got)):
        torch.manual_seed(77)      # RandomCrop3D draws from the global generator (num_workers = 0: this process)
        loaders, _ = make_dataloaders(split, tree / "10" / "hr_is_in_build.npy", batch_size=2, num_workers=0, seed=3,
                                      device_pipeline=pipe, **kw)
        for kind in ("train", "valid"):
            for Xs, bs, ys in loaders[kind]:       # the generator seeds shuffling and crops identically
                sink.append((Xs.cpu(), bs.cpu(), ys.cpu()))
    assert len(got) == len(ref) > 0
    for a, b in zip(got, ref):
        for u, v in zip(a, b):
            assert u.shape == v.shape and torch.equal(u, v)


def test_preprocess_special_values(eng):
    x = torch.tensor([float("nan"), float("inf"), -float("inf"), 310.0, 250.0, 302.0]).view(1, 1, 1, 1, 6)
    out = eng.ops.preprocess(x.to(DEV), [302.0], [8.4], None, True, -7.0, 0).cpu()
    # (tensor-valued mean / std as in the Dataset: with a python scalar torch multiplies by the reciprocal instead)
    m1, s1 = torch.tensor([302.0]).view(1, 1, 1, 1, 1), torch.tensor([8.4]).view(1, 1, 1, 1, 1)
    ref = torch.nan_to_num(torch.clamp((x - m1) / s1, 0.0, 1.0), nan=-7.0)
    assert torch.equal(out, ref)
    out = eng.ops.preprocess(x.to(DEV), [302.0], [8.4], 0.9, False, 0.5, 0).cpu()
    ref = torch.nan_to_num((0.9 * x - m1) / s1, nan=0.5)
    assert torch.equal(out, ref)
    x2 = torch.rand(2, 4, 8, 2, 3) * 40 + 280
    out = eng.ops.preprocess(x2.to(DEV), [300.0, 0.0, 1.0, 2.0], [8.0, 14.0, 21.0, 7.0], None, True, 0.25, 2).cpu()
    m = torch.tensor([300.0, 0.0, 1.0, 2.0]).view(1, 4, 1, 1, 1)
    sd = torch.tensor([8.0, 14.0, 21.0, 7.0]).view(1, 4, 1, 1, 1)
    ref = torch.clamp((x2 - m) / sd, 0.0, 1.0)
    ref[:, :, :2] = 0.25
    assert torch.equal(out, ref)

"""The N > 1 code paths that no single-GPU box can run, rehearsed on two CPU ranks over gloo:

 * bench.py's own ``measure(..., use_dist=True)``: parameter broadcast, the step closure (forward, loss, zero_grad,
   backward with the bucketed all-reduce fired from the autograd hooks, ``reducer.finish()`` BEFORE the optimizer step,
   1/world folded into Adam), barriers around the timed region, MAX all-reduce of the elapsed time;
 * script/train_model.py's per-rank function ``train_and_validate`` launched with ``mp.spawn`` exactly as its ``main``
   does: process group, DistributedSampler loaders, ``train_ddp`` / ``test_ddp`` with their per-epoch SUM all-reduce,
   the three barriers per epoch, rank-0 checkpoint + history, GradNorm variant included.

The HIP engine has no CPU path, so the arithmetic inside the step is a stub (tests/cpu_engine_stub.py: a two-layer ATen
model, a CPU twin of FlatAdam on the real flat layout); GradAllReducer, GradNorm, the loops, the loaders and the two
drivers are the real code.  reference: pytorch/script/train_model.py:104-105,179,212,225,237,336-341;
pytorch/src/optim_helper.py:137-225."""
import importlib.util
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ------------------------------------------------------------------------------------------------ bench.measure
LR_GRID, BATCH, STEPS, WARMUP = [2, 4, 4], 2, 3, 1


def _bench_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import cpu_engine_stub as stub
    m = bench.measure(stub, stub.L, torch.device("cpu"), rank, world, True, BATCH, "l1", LR_GRID, STEPS, WARMUP,
                      breakdown_steps=1)
    flat = stub.LAST_OPT.flat_param.clone()
    got = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(got, flat)
    # numpy arrays travel through the queue BY VALUE (pickled); a torch tensor would travel as a shared-memory handle
    # that the parent can only open while this process is still alive -- an EOFError once in a few runs
    q.put((rank, m["elapsed"], m["voxels_per_step"], m["loss"], [g.numpy().copy() for g in got]))
    dist.destroy_process_group()


def _replay_without_communication(world):
    """what DDP semantics prescribe: every step, the mean over ranks of the per-rank gradients, then one Adam step"""
    import bench
    import cpu_engine_stub as stub
    cfg = bench.make_config("l1")
    scale = 2 ** cfg["model"]["num_x2upsample"]
    hr = tuple(v * scale for v in LR_GRID)
    torch.manual_seed(42)
    model = stub.make_model(cfg)
    loss_fn = stub.make_loss(cfg)
    opt = stub.FlatAdam(model.parameters(), lr=cfg["train"]["lr"])
    data = [bench.synthetic_batch(BATCH, hr, scale, 1234 + r, "cpu") for r in range(world)]
    for _ in range(WARMUP + STEPS + 1):          # warm-up + timed + the one breakdown step
        acc = torch.zeros_like(opt.flat_grad)
        for x, b, y in data:
            opt.zero_grad()
            loss_fn(model(x, b), y, b).backward()
            acc += opt.flat_grad
        opt.flat_grad.copy_(acc / world)
        opt.step()
    return opt.flat_param.clone()


def test_bench_measure_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, e0, v0, l0, g0), (_, e1, v1, l1, g1) = out
    g0, g1 = [torch.from_numpy(a) for a in g0], [torch.from_numpy(a) for a in g1]
    assert e0 == e1 and e0 > 0                               # MAX over ranks: both report the same elapsed time
    hr = [4 * v for v in LR_GRID]
    assert v0 == v1 == world * BATCH * hr[0] * hr[1] * hr[2]    # whole-job voxels per step
    assert l0 != l1                                          # different data per rank (seed 1234 + rank) ...
    assert torch.equal(g0[0], g0[1]) and torch.equal(g0[0], g1[0])   # ... and identical parameters after training
    ref = _replay_without_communication(world)
    assert torch.allclose(g0[0], ref, rtol=1e-5, atol=1e-7), float((g0[0] - ref).abs().max())


# ------------------------------------------------------------------------------------------------ graph + reducer
def _graph_worker(rank, world, port, q):
    """GraphedTrainStep WITH the gradient reducer (BASELINE configs[4]: "8 GPUs and a captured step"), both ways of
    composing them, against the eager step with the reducer -- on CPU ranks a "graph" is a recorded callable, so what is
    rehearsed is the control flow: muted hooks inside segment A, the order and number of collectives, warm-up undone"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import cpu_engine_stub as stub
    import sr3d_amd
    cfg = bench.make_config("l1")
    hr, scale = (8, 16, 16), 4
    batches = [bench.synthetic_batch(2, hr, scale, 100 * rank + i, "cpu") for i in range(3)]

    def run(mode):
        torch.manual_seed(42 + (rank if mode == "unsynced_init" else 0))
        model = stub.make_model(cfg)
        loss_fn = stub.make_loss(cfg)
        opt = stub.FlatAdam(model.parameters(), lr=1e-3, capturable=True)
        red = stub.GradAllReducer(opt.params, opt.flat_grad, opt.offsets, bucket_bytes=1024)
        red.broadcast_parameters(opt.flat_param)
        calls = {"n": 0}
        launch = red._launch

        def counted(b):
            calls["n"] += 1
            launch(b)
        red._launch = counted
        losses = []
        if mode == "eager":
            for x, b, y in batches:
                loss = loss_fn(model(x, b), y, b)
                opt.zero_grad()
                loss.backward()
                opt.grad_scale = red.finish()
                opt.step()
                losses.append(float(loss.detach()))
        else:
            g = sr3d_amd.GraphedTrainStep(model, loss_fn, opt, *batches[0], reducer=red, comm=mode)
            assert opt._host_step == 0                    # the warm-up inside the constructor left no trace
            calls["n"] = 0
            losses = [float(g(x, b, y)) for x, b, y in batches]
        red.remove_hooks()
        return losses, opt.flat_param.clone().numpy(), calls["n"], len(red.buckets)

    out = {m: run(m) for m in ("eager", "split", "captured")}
    q.put((rank, out))
    dist.destroy_process_group()


def test_graphed_step_with_reducer_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        e_loss, e_par, e_calls, nb = out[r]["eager"]
        assert nb >= 2 and e_calls == 3 * nb                 # one all-reduce per bucket and step
        for mode in ("split", "captured"):
            loss, par, calls, _ = out[r][mode]
            assert loss == e_loss, (mode, loss, e_loss)     # same step, same collectives: bit-equal on every rank
            assert np.array_equal(par, e_par), mode
            assert calls == 3 * nb, (mode, calls)
    assert np.array_equal(out[0]["split"][1], out[1]["split"][1])      # ranks in lockstep
    assert out[0]["eager"][0] != out[1]["eager"][0]                     # ... on different data


# ------------------------------------------------------------------------------------------------ train_model.py
def _load_train_model():
    spec = importlib.util.spec_from_file_location(
        "sr3d_train_model", os.path.join(ROOT, "3d-sr-micrometeorology_amd", "script", "train_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _train_worker(rank, world, config, weight_path, history_path, data_root, port):
    """what train_model.main() hands to mp.spawn, with the GPU-only factories swapped for the CPU stub"""
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    os.environ["SR3D_DIST_BACKEND"] = "gloo"
    import cpu_engine_stub as stub
    import sr3d_amd
    sr3d_amd.make_model, sr3d_amd.make_loss, sr3d_amd.FlatAdam = stub.make_model, stub.make_loss, stub.FlatAdam
    tm = _load_train_model()
    tm.train_and_validate(rank, world, config, weight_path, history_path, data_root, port)
    torch.save(stub.LAST_OPT.flat_param, os.path.join(os.path.dirname(weight_path), f"flat_rank{rank}.pt"))


@pytest.mark.parametrize("variant", ["plain", "gradnorm", "relative_result_root", "hip_graph"])
def test_train_and_validate_two_ranks_gloo(tmp_path, variant, monkeypatch):
    from data_fixture import write_synthetic_tree
    data_root = write_synthetic_tree(tmp_path / "d", HR=(8, 16, 16), days=6)
    config = {
        "data": {"data_dir_names": ["10"], "train_valid_test_ratios": [0.6, 0.2, 0.2], "hr_org_size": [8, 16, 16],
                 "hr_crop_size": [8, 8, 16], "means": [302.0, -6.5, -9.1, -3.5], "stds": [8.4, 14.4, 21.6, 7.0],
                 "datasizes": {"train": 100, "valid": 100, "test": 100}, "nan_value": 0.0, "batch_size": 4, "seed": 42,
                 "num_workers": 0},
        "train": {"num_epochs": 2, "lr": 1.0e-3, "num_loops_train": 1, "num_loops_valid": 1, "seed": 42,
                  "loss": {"name": "L1"}},
        "model": {"model_name": "unet", "num_x2upsample": 2},
    }
    if variant == "gradnorm":
        config["train"]["grad_norm"] = {"n_tasks": 3, "alpha": 1.5, "lr": 1.0e-2}
    if variant == "hip_graph":       # the captured step WITH the reducer (split form), inside the reference's epoch loops
        config["train"]["hip_graph"] = True
    res = tmp_path / "res"
    res.mkdir()
    (res / "config.yml").write_text(yaml.safe_dump(config))
    weight_path, history_path, rendezvous = str(res / "weights.pth"), str(res / "learning_history.csv"), str(res / ".rendezvous")
    if variant == "relative_result_root":     # `--result_root out`: "file://res/.rendezvous" would be host "res", path "/.rendezvous"
        monkeypatch.chdir(tmp_path)
        weight_path, history_path, rendezvous = "res/weights.pth", "res/learning_history.csv", "res/.rendezvous"
    world = 2
    mp.spawn(_train_worker, args=(world, config, weight_path, history_path, str(data_root), rendezvous),
             nprocs=world, join=True)      # raises if a rank fails or hangs up (file rendezvous, as train_model.main uses)
    sd = torch.load(weight_path)
    assert set(sd) == {"body.weight", "body.bias", "last.weight", "last.bias"}
    hist = open(history_path).read().strip().splitlines()
    assert hist[0] == "loss,val_loss" and len(hist) == 3
    flats = [torch.load(res / f"flat_rank{r}.pt") for r in range(world)]
    assert torch.equal(flats[0], flats[1])                   # broadcast + averaged gradients: ranks stay in lockstep
    assert "Epoch: 2" in (res / "log.txt").read_text()
    assert (res / "grad_norm_weights_cpu.csv").exists() == (variant == "gradnorm")
    assert ("training step captured into a hipGraph" in (res / "log.txt").read_text()) == (variant == "hip_graph")
    if variant == "hip_graph":
        assert "gradient averaging: split" in (res / "log.txt").read_text()

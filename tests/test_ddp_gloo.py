"""Data-parallel gradient averaging (GradAllReducer) on 2 CPU processes over gloo.

The reducer is device-agnostic host logic: here it runs on CPU tensors; on the
GPU box the same code drives RCCL (backend "nccl")."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flatten(params):
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += (p.numel() + 3) // 4 * 4
    flat = torch.zeros(off)
    for p, o in zip(params, offsets):
        p.grad = flat[o:o + p.numel()].view_as(p)
    return flat, offsets


def _worker(rank, world, port, bucket_bytes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sr3d_amd
    torch.manual_seed(0)  # same weights on every rank
    net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.Tanh(), torch.nn.Linear(33, 5), torch.nn.Tanh(),
                              torch.nn.Linear(5, 3))
    params = list(net.parameters())
    flat, offsets = _flatten(params)
    red = sr3d_amd.GradAllReducer(params, flat, offsets, bucket_bytes=bucket_bytes)
    results = []
    for step in range(2):  # two steps: the reducer must re-arm itself
        flat.zero_()
        g = torch.Generator().manual_seed(100 * step + rank)  # different data per rank
        x = torch.rand(4, 7, generator=g)
        net(x).square().sum().backward()
        scale = red.finish()
        results.append((flat * scale).clone())
    # reference: average of the per-rank gradients computed locally without communication
    ref = []
    for step in range(2):
        acc = torch.zeros_like(flat)
        for r in range(world):
            net2 = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.Tanh(), torch.nn.Linear(33, 5),
                                       torch.nn.Tanh(), torch.nn.Linear(5, 3))
            net2.load_state_dict(net.state_dict())
            g = torch.Generator().manual_seed(100 * step + r)
            x = torch.rand(4, 7, generator=g)
            net2(x).square().sum().backward()
            for p, o in zip(net2.parameters(), offsets):
                acc[o:o + p.numel()] += p.grad.flatten()
        ref.append(acc / world)
    ok = all(torch.allclose(a, b, rtol=1e-6, atol=1e-7) for a, b in zip(results, ref))
    q.put((rank, ok, len(red.buckets)))
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes,min_buckets", [(64 << 20, 1), (256, 3)])
def test_bucketed_allreduce_matches_mean_of_rank_gradients(bucket_bytes, min_buckets):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_bytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in out), out
    assert all(nb >= min_buckets for _, _, nb in out), out


# ---------------------------------------------------------------------------------------------------------------
# the REAL model's layout: UNetSR at default.yml widths (48 tensors, 65.47 M parameters = 262 MB), laid out by the
# same flatten_parameters() FlatAdam uses, default 64 MB buckets (5 of them) in reverse parameter order


def _real_layout_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import sr3d_amd
    from sr3d_amd.src.optim import flatten_parameters
    torch.manual_seed(1)
    model = sr3d_amd.make_model(bench.make_config("l1"))       # CPU tensors: only the host-side layout is used
    params, offsets, flat_param, flat_grad = flatten_parameters(model.parameters())
    red = sr3d_amd.GradAllReducer(params, flat_grad, offsets)
    if rank != 0:
        flat_param.zero_()
    red.broadcast_parameters(flat_param)                       # DDP's initial weight sync
    checks = {"n_params": len(params), "numel": int(flat_param.numel()), "n_buckets": len(red.buckets),
              "bucket_mb": [round((b["end"] - b["begin"]) * 4 / 2 ** 20, 1) for b in red.buckets],
              "covers": sorted((b["begin"], b["end"]) for b in red.buckets)}
    torch.manual_seed(1)
    ref_model = sr3d_amd.make_model(bench.make_config("l1"))
    checks["broadcast_ok"] = all(torch.equal(a, b) for a, b in zip(model.parameters(), ref_model.parameters()))
    del ref_model
    ok = True
    for step in range(2):
        flat_grad.zero_()
        # a loss whose gradient w.r.t. parameter p is a known rank-dependent field c_{rank,step} * (1 + index mod 7):
        # autograd runs the 48 post-accumulate hooks, the buckets fire as their last member arrives
        coef = 1.0 + rank + 10.0 * step
        loss = sum((p * _field(p, coef)).sum() for p in params)
        loss.backward()
        scale = red.finish()
        mean_coef = sum(1.0 + r + 10.0 * step for r in range(world)) / world
        for p, o in zip(params, offsets):
            got = flat_grad[o:o + p.numel()] * scale
            ok = ok and torch.allclose(got, _field(p, mean_coef).flatten(), rtol=1e-6, atol=0)
        # padding between parameters must stay zero (it is all-reduced with the buckets)
        pad = flat_grad.clone()
        for p, o in zip(params, offsets):
            pad[o:o + p.numel()] = 0
        ok = ok and float(pad.abs().max()) == 0.0
    checks["grads_ok"] = ok
    q.put((rank, checks))
    dist.destroy_process_group()


def _field(p, coef):
    idx = torch.arange(p.numel(), dtype=torch.float32) % 7
    return (coef * (1.0 + idx)).view_as(p)


def test_real_model_layout_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_layout_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, c in out.items():
        assert c["n_params"] == 48 and c["numel"] >= 65_472_736, c
        assert c["n_buckets"] == 5, c                          # 262 MB in 64 MB buckets
        # buckets tile the flat buffer without gaps or overlap
        cov = c["covers"]
        assert cov[0][0] == 0 and cov[-1][1] == c["numel"]
        assert all(a[1] == b[0] for a, b in zip(cov, cov[1:])), cov
        assert c["broadcast_ok"] and c["grads_ok"], (rank, c)

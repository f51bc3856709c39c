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

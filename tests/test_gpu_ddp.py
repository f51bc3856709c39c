"""Data-parallel path on the GPU box (one MI355X): RCCL with a single rank (initialisation, broadcast, bucketed
all-reduce on the side stream, Adam with the folded 1/world factor), and the engine's 2-rank gradient equality --
two processes sharing cuda:0, each with half of a global batch, against one process with the whole batch
(SURVEY.md section 4 item 4).  Two ranks cannot share one GPU under RCCL, so that test moves the buckets with the
gloo backend; the reducer code (hooks, buckets, side stream, finish) is the same.  No scaling number comes out of
this file: it checks values only."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import T, cfg_of, load_golden, relerr, sub

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _l1_cfg(d):
    cfg = cfg_of(d)
    cfg["train"]["loss"] = {"name": "L1"}
    return cfg


def _step(eng, cfg, sd, x, b, y, reducer_factory=None, bucket_bytes=None):
    model = eng.make_model(cfg)
    model.load_state_dict(sd)
    model.to(DEV)
    opt = eng.FlatAdam(model.parameters(), lr=1e-3)
    red = None
    if reducer_factory is not None:
        red = reducer_factory(opt)
        red.broadcast_parameters(opt.flat_param)
    loss = eng.make_loss(cfg)(model(x.to(DEV), b.to(DEV)), y.to(DEV), b.to(DEV))
    opt.zero_grad()
    loss.backward()
    if red is not None:
        opt.grad_scale = red.finish()
    grads = (opt.flat_grad * opt.grad_scale).clone()
    opt.step()
    torch.cuda.synchronize()
    if red is not None:
        red.remove_hooks()
    return float(loss.detach()), grads.cpu(), opt.flat_param.detach().clone().cpu(), (len(red.buckets) if red else 0)


def _rccl_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))     # nccl == RCCL on ROCm
    import sr3d_amd as eng
    d = load_golden("model_tiny_a.npz")
    cfg, sd = _l1_cfg(d), sub(d, "sd")
    x, b, y = T(d["x"]), T(d["b"]), T(d["y"])
    plain = _step(eng, cfg, sd, x, b, y)
    # 4 KB buckets: every parameter its own bucket -> 46 asynchronous all-reduces on the side stream
    ddp = _step(eng, cfg, sd, x, b, y, lambda opt: eng.GradAllReducer(opt.params, opt.flat_grad, opt.offsets,
                                                                      bucket_bytes=4096))
    q.put({"loss_equal": plain[0] == ddp[0], "grads_equal": bool(torch.equal(plain[1], ddp[1])),
           "params_equal": bool(torch.equal(plain[2], ddp[2])), "buckets": ddp[3],
           "backend": dist.get_backend()})
    dist.destroy_process_group()


def test_single_rank_rccl_walks_the_whole_ddp_path():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert out["backend"] == "nccl" and out["buckets"] >= 40, out
    assert out["loss_equal"] and out["grads_equal"] and out["params_equal"], out


def _rccl_graph_worker(port, q, modes):
    """the captured step WITH the reducer over RCCL (one rank): `split` = graph A (forward .. backward, hooks muted) + eager
    bucket all-reduces + Adam; `captured` = the all-reduces captured on the reducer's side stream inside the one graph"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    import sr3d_amd as eng
    from helpers import synthetic_inputs
    d = load_golden("model_tiny_a.npz")
    cfg, sd = _l1_cfg(d), sub(d, "sd")
    batches = [tuple(t.to(DEV) for t in synthetic_inputs(2, (16, 16, 16), 4, 50 + i, "iid")) for i in range(3)]
    res = {}
    for mode in modes:
        model = eng.make_model(cfg)
        model.load_state_dict(sd)
        model.to(DEV)
        loss_fn = eng.make_loss(cfg)
        opt = eng.FlatAdam(model.parameters(), lr=1e-3, capturable=mode != "eager")
        red = eng.GradAllReducer(opt.params, opt.flat_grad, opt.offsets, bucket_bytes=4096)
        red.broadcast_parameters(opt.flat_param)
        try:
            if mode == "eager":
                losses = []
                for x, b, y in batches:
                    loss = loss_fn(model(x, b), y, b)
                    opt.zero_grad()
                    loss.backward()
                    opt.grad_scale = red.finish()
                    opt.step()
                    losses.append(float(loss.detach()))
            else:
                g = eng.GraphedTrainStep(model, loss_fn, opt, *batches[0], reducer=red, comm=mode)
                losses = [float(g(*bt)) for bt in batches]
            torch.cuda.synchronize()
            res[mode] = {"losses": losses, "param": opt.flat_param.detach().cpu().numpy(), "buckets": len(red.buckets)}
        except Exception as e:      # noqa: BLE001  (reported to the parent, which asserts)
            res[mode] = {"error": f"{type(e).__name__}: {e}"[:400]}
        red.remove_hooks()
    q.put(res)
    dist.destroy_process_group()


def _run_graph_worker(modes, limit_s):
    """the worker in its own process with a HARD limit: a hung collective must cost this test, not the whole run (the GPU box
    kills a command that stays silent for 7 minutes); returns None when the limit was hit (the process is killed by its PID)"""
    import queue as _queue
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_graph_worker, args=(_free_port(), q, modes))
    p.start()
    try:
        out = q.get(timeout=limit_s)
    except _queue.Empty:
        out = None
    p.join(timeout=30 if out is not None else 1)
    if p.is_alive():
        p.kill()
        p.join(timeout=30)
    elif out is not None:
        assert p.exitcode == 0
    return out


def test_single_rank_rccl_graphed_step_with_reducer_equals_eager_step_with_reducer():
    """the DEFAULT composition of the captured step and the reducer (`split`: graph A + eager bucket all-reduces + Adam)"""
    import numpy as np
    out = _run_graph_worker(("eager", "split"), 240)
    assert out is not None, "the worker did not finish within 240 s"
    assert "error" not in out["eager"], out["eager"]
    assert out["eager"]["buckets"] >= 40
    assert "error" not in out["split"], out["split"]
    assert out["split"]["losses"] == out["eager"]["losses"]
    assert np.array_equal(out["split"]["param"], out["eager"]["param"])


def test_single_rank_rccl_captured_inside_the_graph_equals_eager_step_with_reducer():
    """the opt-in composition (`SR3D_GRAPH_COMM=captured`): RCCL's all-reduces captured inside the graph.  Held to the same
    bit-equality where this runtime captures them; a capture that RCCL refuses or that does not come back within the limit is
    reported as an expected failure of the OPTION (the default, `split`, is the test above)"""
    import numpy as np
    out = _run_graph_worker(("eager", "captured"), 150)
    if out is None:
        pytest.xfail("RCCL captured inside a hipGraph did not come back within 150 s on this box (opt-in mode)")
    if "error" in out["captured"]:
        pytest.xfail("RCCL refused the capture: " + out["captured"]["error"])
    assert out["captured"]["losses"] == out["eager"]["losses"]
    assert np.array_equal(out["captured"]["param"], out["eager"]["param"])


def _two_rank_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sr3d_amd as eng
    d = load_golden("model_tiny_a.npz")                       # global batch of 2 samples
    cfg, sd = _l1_cfg(d), sub(d, "sd")
    x, b, y = T(d["x"]), T(d["b"]), T(d["y"])
    if rank != 0:                                             # the broadcast must repair this
        sd = {k: torch.zeros_like(v) for k, v in sd.items()}
    out = _step(eng, cfg, sd, x[rank:rank + 1], b[rank:rank + 1], y[rank:rank + 1],
                lambda opt: eng.GradAllReducer(opt.params, opt.flat_grad, opt.offsets, bucket_bytes=64 << 10))
    q.put((rank, (out[0], out[1].numpy(), out[2].numpy(), out[3])))   # numpy: pickled by value, no fd passing
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank_on_the_same_global_batch():
    import sr3d_amd as eng
    d = load_golden("model_tiny_a.npz")
    cfg, sd = _l1_cfg(d), sub(d, "sd")
    whole = _step(eng, cfg, sd, T(d["x"]), T(d["b"]), T(d["y"]))          # one process, batch 2
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    out = {r: (o[0], torch.from_numpy(o[1]), torch.from_numpy(o[2]), o[3]) for r, o in out.items()}
    # mean |p - t| over 2 samples = mean of the per-sample means: averaged rank gradients = whole-batch gradient
    assert abs(0.5 * (out[0][0] + out[1][0]) - whole[0]) < 1e-6 * whole[0]
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])     # ranks agree bit for bit
    assert relerr(out[0][1], whole[1]) < 1e-5
    assert relerr(out[0][2], whole[2]) < 1e-5               # parameters after the Adam step
    assert out[0][3] > 5

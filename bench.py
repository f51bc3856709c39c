#!/usr/bin/env python3
"""Training-step throughput of the voxel-SR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = forward + loss + zero_grad + backward (+ gradient all-reduce for
N > 1) + Adam of the reference's UNetSR (default.yml widths, 65.47 M parameters)
on one synthetic batch, i.e. the body of reference
pytorch/src/optim_helper.py:156-178.

Workload (HR 80x320x320 from LR 20x80x80, fp32, weak scaling):
  N = 1 : BASELINE.json configs[1] -- batch 1, L1 loss.  A short second measurement of configs[2] (batch 4, the
          physics-guided loss: the per-GPU work of configs[3]) is attached as "config2_batch4_mixed".
  N > 1 : BASELINE.json configs[3] -- batch 4 per GPU (global 4 N), physics-guided loss, gradient all-reduce over
          RCCL.  A short second measurement of configs[1]'s per-GPU work (batch 1, L1) is attached as
          "config1_batch1_l1", so that per-GPU throughput can be compared with the N = 1 line like for like.
``--batch`` / ``--loss`` override the primary workload.

Prints ONE JSON line (rank 0) with the metric, the roofline figure of the
dominant kernel measured live with HIP events, GB/s of the HBM-bound kernel families,
and a CPU baseline (the oracle, timed on this host's cores on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic work per HR voxel per training step, default.yml widths
FLOP_PER_VOXEL = 8_486_693
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/F16 ~2.5 PF dense (v_mfma_f32_32x32x16_f16: 32 cycles per SIMD)
# what tools/mfma_rate.hip sustains for 100 ms with register operands only: the f16 MFMA loop runs into the power limit.
# v_mfma_f32_16x16x32_f16 (the shape hconv_kernel uses): 3 x 634.6 TFLOP/s at 1.89 GHz; 32x32x16: 3 x 558 at 1.70 GHz
# (fp32 MFMA: 152 TFLOP/s at 2.33 GHz, not throttled)
F16_MFMA_SUSTAINED_TFLOPS = 1904.0
# HBM traffic of the dominant kernel family per launch, from separate rocprofv3 --pmc passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass; FETCH_SIZE doubled as the guide prescribes for gfx950)
TRAFFIC_JSONS = [os.path.join(ROOT, "profiles", n) for n in ("r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02b_pmc_hbm_traffic.json")]
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured with a float4 copy)
BYTES_PER_VOXEL = 11_586       # compulsory fwd+bwd activation traffic per HR voxel, fp32
FAMILIES = {"igemm_s1": 0, "igemm_s2": 1, "igemm_bwd_s2": 2, "wgrad": 3, "loss": 4, "act_bwd": 5, "bias_grad": 6,
            "adam": 7, "data": 8, "pack_reduce": 9, "hconv": 11, "hconv_small": 12}

DEFAULT_CONFIG = {
    "data": {"stds": [8.40, 14.40, 21.60, 7.00]},
    "train": {"lr": 1.0e-4, "loss": {"name": "L1"}},
    "model": {"model_name": "unet", "in_channels": 4, "out_channels": 4, "num_feat0": 64, "num_feat1": 128,
              "num_feat2": 128, "num_feat3": 256, "num_feat4": 256, "num_x2upsample": 2, "num_latent_layers": 3,
              "n_layers_in_block": 2, "bias_feat_extraction": False,
              "conv_mode_feat_extraction": "g_conv_with_separated_bias",
              "conv_mode_down_block": "g_conv_with_separated_bias", "conv_mode_up_block": None},
}


def make_config(loss: str) -> dict:
    cfg = json.loads(json.dumps(DEFAULT_CONFIG))
    if loss == "mixed":
        cfg["train"]["loss"] = {"name": "MixedDivergenceGradientL2Loss", "weight_gradient_loss": 1.0,
                                "weight_divergence_loss": 10.0}
    return cfg


def synthetic_batch(batch, hr, scale, seed, device):
    g = torch.Generator().manual_seed(seed)
    Z, Y, X = hr
    x = torch.rand(batch, 4, Z // scale, Y // scale, X // scale, generator=g)
    y = torch.rand(batch, 4, Z, Y, X, generator=g)
    b = (torch.rand(batch, 1, Z, Y, X, generator=g) > 0.2).float()
    return x.to(device), b.to(device), y.to(device)


def host_cores() -> int:
    """cores this process may really use: the GPU box gives one GPU's share (16) of a 256-thread host"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, budget_s=20.0):
    """the CPU oracle (oracle/ref_cpu.py: stock ATen conv3d etc., pinned to the reference by
    tests/golden) on the reference's own training crop size, timed on this host's cores"""
    from oracle import ref_cpu as R
    cores = host_cores()
    torch.set_num_threads(cores)
    # the GPU workload's own model (4x, default.yml widths) and loss on a volume the CPU finishes in about a second:
    # HR 32x64x64 (the HR volume of BASELINE.json configs[0], and default.yml's training crop) from LR 8x16x16
    cfg = json.loads(json.dumps(cfg))
    scale = 2 ** cfg["model"]["num_x2upsample"]
    hr = (32, 64, 64)
    sd = R.random_state_dict(cfg["model"], seed=42)
    opt = R.AdamState(sd, lr=cfg["train"]["lr"])
    x, b, y = synthetic_batch(1, hr, scale, 1234, "cpu")
    R.train_step(sd, opt, cfg, x, b, y)  # warm-up (first call pays oneDNN primitive creation)
    times = []
    t_all = time.time()
    while True:
        t0 = time.time()
        R.train_step(sd, opt, cfg, x, b, y)
        times.append(time.time() - t0)
        if len(times) >= 5 or time.time() - t_all + times[-1] > budget_s:
            break
    sec = sum(times) / len(times)
    vox = hr[0] * hr[1] * hr[2]
    return {"value": vox / sec, "unit": "HR voxels/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{len(times)} full training steps (fwd+loss+bwd+Adam) of the CPU oracle: the SAME model ({scale}x SR, "
                      f"default.yml widths), loss and optimizer as the GPU line, on a smaller volume -- LR "
                      f"{hr[0] // scale}x{hr[1] // scale}x{hr[2] // scale} -> HR {hr[0]}x{hr[1]}x{hr[2]} (131,072 HR voxels, the HR "
                      f"volume of BASELINE configs[0] and default.yml's training crop), batch 1, {sec:.2f} s/step; the "
                      f"80x320x320 volume of the GPU workload needs ~35 GB and minutes per step on CPU"}


def read_profile(L):
    prof = {}
    for name, kid in list(FAMILIES.items()) + [("dropped", 99)]:
        ms, work, n = C.c_double(), C.c_double(), C.c_longlong()
        L.check(L.lib.sr3d_profile_read(kid, C.byref(ms), C.byref(work), C.byref(n)), "sr3d_profile_read")
        prof[name] = {"ms": ms.value, "work": work.value, "launches": n.value}
    return prof


def measure(sr3d_amd, L, dev, rank, world, use_dist, batch, loss_name, lr_grid, steps, warmup, graph=False, breakdown_steps=0,
            storage="fp32"):
    """W untimed + K timed training steps of one workload; returns wall time (max over ranks), the HIP-event totals of
    the dominant (stride-1 convolution) kernel family inside the timed steps, and -- from `breakdown_steps` further,
    UNTIMED steps -- the totals of every family.  (Inside the timed region only the dominant family is bracketed: 48
    instead of ~1100 event records per step; measured cost of bracketing everything: 0.3-0.7 % of the step.)"""
    import gc
    gc.collect()              # a previous leg's hipGraph (and its private memory pool) dies with its last reference
    if torch.device(dev).type == "cuda":
        torch.cuda.empty_cache()
    cfg = make_config(loss_name)
    if storage != "fp32":     # engine extension: activations and their gradients stored as bf16 (BASELINE configs[4])
        cfg["model"]["storage_dtype"] = storage
    scale = 2 ** cfg["model"]["num_x2upsample"]
    hr = tuple(v * scale for v in lr_grid)
    torch.manual_seed(42)
    model = sr3d_amd.make_model(cfg).to(dev)
    loss_fn = sr3d_amd.make_loss(cfg)
    opt = sr3d_amd.FlatAdam(model.parameters(), lr=cfg["train"]["lr"], capturable=graph)
    reducer = None
    if use_dist:
        reducer = sr3d_amd.GradAllReducer(opt.params, opt.flat_grad, opt.offsets)
        reducer.broadcast_parameters(opt.flat_param)
    x, b, y = synthetic_batch(batch, hr, scale, 1234 + rank, dev)
    if torch.device(dev).type == "cuda":
        torch.cuda.reset_peak_memory_stats(dev)

    def step():
        pred = model(x, b)
        loss = loss_fn(pred, y, b)
        opt.zero_grad()
        loss.backward()
        if reducer is not None:
            opt.grad_scale = reducer.finish()
        opt.step()
        return loss

    if graph:      # the whole step captured once into a hipGraph; every step below is one replay (+ with N > 1 the bucket
        #            all-reduces: eager between the captured backward and Adam, or captured too -- src/graph.py)
        gstep = sr3d_amd.GraphedTrainStep(model, loss_fn, opt, x, b, y, reducer=reducer)

        def step():  # noqa: F811
            return gstep(x, b, y)

    on_gpu = torch.device(dev).type == "cuda"      # (tests/test_dist_paths_gloo.py drives this function on CPU ranks)

    def fence():
        if on_gpu:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    losses = []                         # the loss of every step (device scalars; read after the timed region)

    def keep(loss):
        losses.append(loss.detach().clone() if graph else loss.detach())

    for _ in range(warmup):
        keep(step())
    fence()
    if not graph:                       # (event records cannot be part of a captured step)
        # dominant family only; the event pool is created here, outside the timed region
        # (SR3D_BENCH_PROFILE_MODE=1 brackets every launch in the timed region, =0 none: for measuring that overhead)
        L.profile_enable(int(os.environ.get("SR3D_BENCH_PROFILE_MODE", "2")))
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
        keep(loss)
    fence()
    elapsed = time.perf_counter() - t0
    last_loss = float(loss.detach())
    losses = [float(v) for v in losses]

    prof = read_profile(L)              # dominant family, timed steps
    L.profile_enable(False)
    breakdown = None
    if breakdown_steps > 0 and not graph:
        L.profile_enable(1)             # every family, untimed
        fence()
        for _ in range(breakdown_steps):
            step()
        fence()
        breakdown = read_profile(L)
        L.profile_enable(False)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if reducer is not None:
        reducer.remove_hooks()
    del model, opt, reducer, x, b, y, loss
    if on_gpu:
        torch.cuda.empty_cache()
    return {"elapsed": float(t.item()), "prof": prof, "breakdown": breakdown, "breakdown_steps": breakdown_steps,
            "loss": last_loss, "losses": losses, "hr": hr, "cfg": cfg, "peak_mem_gb": (torch.cuda.max_memory_allocated(dev) / 2 ** 30 if on_gpu else 0.0),
            "voxels_per_step": world * batch * hr[0] * hr[1] * hr[2]}


def workload_name(lr_grid, hr, batch, loss_name, world, storage="fp32"):
    cfg_id = {("l1", False): 1, ("mixed", False): 2, ("mixed", True): 3}.get((loss_name, world > 1))
    tag = f" (BASELINE configs[{cfg_id}])" if cfg_id is not None and (batch == (1 if loss_name == "l1" else 4)) else ""
    if storage != "fp32" or list(lr_grid) != [20, 80, 80]:
        tag = " (BASELINE configs[4]: per-GPU work)" if (storage == "bf16" and list(lr_grid) == [40, 160, 160] and batch == 1) else ""
    prec = "fp32" if storage == "fp32" else "bf16 storage + bf16 MFMA, fp32 accumulate / master weights / Adam"
    return (f"LR {lr_grid[0]}x{lr_grid[1]}x{lr_grid[2]} -> 4x SR HR {hr[0]}x{hr[1]}x{hr[2]}, batch {batch}/GPU, {prec}, "
            f"UNetSR default.yml widths (65.47M params), {'L1' if loss_name == 'l1' else 'MixedDivergenceGradientL2'} "
            f"loss, fwd+loss+bwd{'+RCCL grad all-reduce' if world > 1 else ''}+Adam{tag}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 1 for N = 1, 4 for N > 1)")
    ap.add_argument("--loss", choices=["l1", "mixed"], default=None, help="default: l1 for N = 1, mixed for N > 1")
    ap.add_argument("--lr-grid", type=int, nargs=3, default=[20, 80, 80], metavar=("Z", "Y", "X"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short second workload")
    ap.add_argument("--storage", choices=["fp32", "bf16"], default="fp32",
                    help="activation storage of the PRIMARY workload (bf16: engine extension, BASELINE configs[4]; for profiling -- "
                         "the default line is fp32, the reference's precision)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the training step into a hipGraph and time replays (no per-kernel breakdown)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and the bucketed all-reduce even with one rank (rehearsal of the N>1 path)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the sr3d engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm

    import sr3d_amd
    from sr3d_amd import _lib as L

    batch = args.batch if args.batch is not None else (1 if world == 1 else 4)
    loss_name = args.loss if args.loss is not None else ("l1" if world == 1 else "mixed")
    m = measure(sr3d_amd, L, dev, rank, world, use_dist, batch, loss_name, args.lr_grid, args.steps, args.warmup,
                graph=args.graph, breakdown_steps=0 if args.graph else 2, storage=args.storage)
    leg_errors = {}

    def attached(name, fn):
        """an attached measurement must never cost the headline line: a failure is reported under `attached_leg_errors`.
        (Single GPU only: with N > 1 every rank must take the same path through the collectives, so errors propagate.)"""
        if use_dist:
            return fn()
        try:
            return fn()
        except Exception as e:   # noqa: BLE001
            leg_errors[name] = f"{type(e).__name__}: {e}"[:300]
            L.profile_enable(False)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            return None

    second = None
    if not args.no_secondary and args.batch is None and args.loss is None and not args.graph:
        sb, sl = (4, "mixed") if world == 1 else (1, "l1")
        second = attached("second", lambda: measure(sr3d_amd, L, dev, rank, world, use_dist, sb, sl, args.lr_grid,
                                                    min(args.steps, 3), 1, breakdown_steps=1))
        if second is not None:
            second["batch"], second["loss_name"] = sb, sl

    fp32_only = None   # the same configuration with every stride-1 layer back on the fp32 Winograd kernel
    if not args.no_secondary and not args.graph and os.environ.get("SR3D_SPLIT_F16", "1") != "0":
        prev = os.environ.get("SR3D_SPLIT_F16")
        os.environ["SR3D_SPLIT_F16"] = "0"
        fp32_only = attached("fp32_mfma_only", lambda: measure(sr3d_amd, L, dev, rank, world, use_dist, batch, loss_name,
                                                               args.lr_grid, args.steps, args.warmup))
        if prev is None:
            del os.environ["SR3D_SPLIT_F16"]
        else:
            os.environ["SR3D_SPLIT_F16"] = prev

    # bf16 storage (engine extension, BASELINE configs[4]): (a) the headline's own shape, eager, with the kernel breakdown --
    # like for like with the fp32 line above; (b) configs[4]'s per-GPU workload: LR 40x160x160 -> HR 160x640x640, batch 1,
    # physics-guided loss, the whole step captured into a hipGraph and replayed
    bf16_same, bf16_c4 = None, None
    if not args.no_secondary and not args.graph and not use_dist and os.environ.get("SR3D_BENCH_BF16", "1") != "0":
        bf16_same = attached("bf16_storage", lambda: measure(sr3d_amd, L, dev, rank, world, use_dist, batch, loss_name, args.lr_grid,
                                                             min(args.steps, 5), 1, breakdown_steps=1, storage="bf16"))
        bf16_c4 = attached("config4_bf16_hipgraph", lambda: measure(sr3d_amd, L, dev, rank, world, use_dist, 1, "mixed", [40, 160, 160],
                                                                    min(args.steps, 3), 1, graph=True, storage="bf16"))

    replay = None      # the same step captured once into a hipGraph and replayed (src/graph.py): no per-launch overhead
    if not args.no_secondary and not args.graph and not use_dist:
        replay = attached("hipgraph_replay", lambda: measure(sr3d_amd, L, dev, rank, world, use_dist, batch, loss_name, args.lr_grid,
                                                             min(args.steps, 5), 2, graph=True))

    if rank == 0:
        elapsed, prof, hr = m["elapsed"], m["prof"], m["hr"]
        value = m["voxels_per_step"] * args.steps / elapsed
        wino = os.environ.get("SR3D_WINOGRAD", "1") != "0"
        split = prof["hconv"]["ms"] > prof["igemm_s1"]["ms"]   # the split-f16 kernel carries the stride-1 layers
        dom_key = "hconv" if split else "igemm_s1"
        dom = prof[dom_key]
        algo = dom["work"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        # MFMA FLOPs the kernel executes per algorithmic FLOP: 3 f16 products per fp32 product (split-f16, direct),
        # 1 / 2.25 (fp32 Winograd F(2x2,3x3) in (y,x)), 1 (direct fp32)
        executed = algo * 3.0 if split else algo / (2.25 if wino else 1.0)
        peak = F16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        traffic = None
        tj = next((f for f in TRAFFIC_JSONS if os.path.exists(f)), None)
        if tj and batch == 1 and loss_name == "l1" and wino:
            t = json.load(open(tj)).get(dom_key)
            if t:
                traffic = {"hbm_bytes_per_launch": t["fetch_bytes_per_launch_x2_gfx950"] + t["write_bytes_per_launch"],
                           "fetch_bytes_per_launch": t["fetch_bytes_per_launch_x2_gfx950"],
                           "write_bytes_per_launch": t["write_bytes_per_launch"],
                           "kernel_launches_profiled": t["launches_profiled"],
                           "source": os.path.relpath(tj, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
        if split:
            kernel_desc = ("stride-1 conv forward + input gradient on hconv_kernel: direct implicit GEMM on "
                           "v_mfma_f32_16x16x32_f16, fp32 operands split into two fp16 halves (3 products), fp32 accumulate")
            note = ("achieved = ALGORITHMIC FLOPs of the 3x3x3 convolution per second (2*27*Cin*Cout per output voxel, SURVEY "
                    "8(d)) / kernel time from HIP events; executed_tflops = 3 x that, the f16 MFMA FLOPs the split scheme "
                    "really issues (channel padding to 16 not counted); over the kernel's launches of >= 448 workgroups (they fill the chip; the launches on "
                    "the small grids of U-Net levels 3-4 are listed as conv_kernels.hconv_small); peak = dense f16 MFMA at 2.4 "
                    "GHz.  The f16 MFMA is POWER-bound on this part: "
                    f"tools/mfma_rate.hip sustains {F16_MFMA_SUSTAINED_TFLOPS:.0f} TFLOP/s with the kernel's MFMA shape, 16x16x32 "
                    "(register operands only, 100 ms, clock settling at 1.89 GHz; 1674 with 32x32x16) -- frac_executed_of_sustained is "
                    "against that")
        else:
            kernel_desc = ("stride-1 conv forward + input gradient (wino_kernel: Winograd F(2x2,3x3) x 3 z-taps on "
                           "v_mfma_f32_32x32x2_f32; direct igemm_kernel with SR3D_WINOGRAD=0)")
            note = ("achieved = ALGORITHMIC FLOPs of the 3x3x3 convolution per second (2*27*Cin*Cout per output voxel, "
                    "SURVEY 8(d)) / kernel time from HIP events; executed_tflops = that / 2.25 (Winograd F(2x2,3x3) in (y,x) "
                    "needs 48 instead of 108 products per 2x2x1 outputs); frac_executed = matrix-pipe utilisation against "
                    "the fp32 MFMA peak (frac can exceed it: Winograd)")
        # every other family: from the untimed breakdown pass (all launches bracketed; ~4 % slower steps)
        bd, bs = m["breakdown"] or prof, (m["breakdown_steps"] if m["breakdown"] else args.steps)
        conv = {k: {"ms_per_step": bd[k]["ms"] / bs, "launches_per_step": bd[k]["launches"] / bs,
                    "algorithmic_tflops": (bd[k]["work"] / (bd[k]["ms"] * 1e-3) / 1e12 if bd[k]["ms"] > 0 else 0.0)}
                for k in ("hconv", "hconv_small", "igemm_s1", "igemm_s2", "igemm_bwd_s2", "wgrad")}
        hbm = {k: {"ms_per_step": bd[k]["ms"] / bs, "launches_per_step": bd[k]["launches"] / bs,
                   "gbytes_per_s": (bd[k]["work"] / (bd[k]["ms"] * 1e-3) / 1e9 if bd[k]["ms"] > 0 else 0.0),
                   "frac_of_hbm_peak": (bd[k]["work"] / (bd[k]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                                        if bd[k]["ms"] > 0 else 0.0)}
               for k in ("loss", "act_bwd", "bias_grad", "adam", "data", "pack_reduce")}
        out = {
            "metric": "training voxels/sec (fwd+bwd+loss) on 4x 3D SR",
            "value": value,
            "unit": "HR voxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("bf16 storage + bf16 MFMA, fp32 accumulate / master weights / Adam" if args.storage == "bf16" else
                      "fp32 (2 x fp16-split MFMA products, fp32 accumulate)" if split else "fp32"),
            "dtype_note": ("storage, accumulation and results are fp32; the products of the chip-filling convolutions (forward, "
                           "input gradient, stride-1 weight gradient) are 3 f16 MFMAs on fp32 operands split exactly into two "
                           "fp16 halves (error 2^-22 per operand; every parity test at 1e-5, measured layer error 4e-7 = that of "
                           "the fp32 MFMA kernels); SR3D_SPLIT_F16=0 computes every product on the fp32 MFMA: fp32_mfma_only"
                           if split else "every product on the fp32 MFMA"),
            "data": "synthetic",
            "config": {"workload": workload_name(args.lr_grid, hr, batch, loss_name, world, args.storage) +
                       (" [hipGraph replay]" if args.graph else ""),
                       "global_batch": world * batch,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "roofline": {"bound": "mfma", "achieved": algo, "peak": peak, "unit": "TFLOP/s",
                         "frac": algo / peak, "frac_algorithmic": algo / peak, "traffic": traffic,
                         "executed_tflops": executed, "frac_executed": executed / peak,
                         "kernel": kernel_desc,
                         "note": note,
                         "algorithmic_tflops": algo,
                         "launches_per_step": dom["launches"] / args.steps,
                         "kernel_ms_per_step": dom["ms"] / args.steps,
                         **({"frac_executed_of_sustained": executed / F16_MFMA_SUSTAINED_TFLOPS} if split else {})},
            "hbm_frac": value / (world * HBM_PEAK_GBS * 1e9 / BYTES_PER_VOXEL),
            "hbm_note": f"voxels/s against the HBM-only ceiling {HBM_PEAK_GBS * 1e9 / BYTES_PER_VOXEL / 1e6:.0f} M voxels/s/GPU "
                        f"(8 TB/s / {BYTES_PER_VOXEL} B of compulsory activation traffic per voxel); the step is "
                        "contraction-bound (AI ~ 730 FLOP/B), so this fraction cannot exceed ~1/16 at fp32-MFMA peak "
                        "even with Winograd",
            "step_algorithmic_tflops": FLOP_PER_VOXEL * value / 1e12,
            "conv_kernels": conv,
            "hbm_bound_kernels": hbm,
            "kernel_breakdown_note": ("conv_kernels / hbm_bound_kernels come from 2 further, UNTIMED steps with every launch "
                                      "bracketed by HIP events; inside the timed region only the roofline kernel family is"),
            "profile_records_dropped": prof["dropped"]["launches"] + (m["breakdown"]["dropped"]["launches"] if m["breakdown"] else 0),
            "loss": m["loss"],
        }
        if second is not None:
            key = "config2_batch4_mixed" if world == 1 else "config1_batch1_l1"
            sv = second["voxels_per_step"] * min(args.steps, 3) / second["elapsed"]
            sp = second["breakdown"]
            out[key] = {"workload": workload_name(args.lr_grid, second["hr"], second["batch"], second["loss_name"], world),
                        "value": sv, "unit": "HR voxels/s", "steps": min(args.steps, 3), "warmup": 1,
                        "ms_per_step": second["elapsed"] / min(args.steps, 3) * 1e3, "loss": second["loss"],
                        "loss_kernels_ms_per_step": sp["loss"]["ms"] / second["breakdown_steps"],
                        "loss_kernels_gbytes_per_s": (sp["loss"]["work"] / (sp["loss"]["ms"] * 1e-3) / 1e9
                                                      if sp["loss"]["ms"] > 0 else 0.0)}
        if fp32_only is not None:
            n5 = args.steps
            fd = fp32_only["prof"]["igemm_s1"]
            out["fp32_mfma_only"] = {"note": "same workload with SR3D_SPLIT_F16=0 (fp32 MFMA kernels only: Winograd stride 1, direct stride 2)",
                                     "value": fp32_only["voxels_per_step"] * n5 / fp32_only["elapsed"], "unit": "HR voxels/s",
                                     "steps": n5, "warmup": args.warmup, "ms_per_step": fp32_only["elapsed"] / n5 * 1e3,
                                     "loss": fp32_only["loss"],
                                     "wino_kernel_frac_of_fp32_mfma_peak": (fd["work"] / (fd["ms"] * 1e-3) / 1e12 / 2.25 /
                                                                            FP32_MFMA_PEAK_TFLOPS if fd["ms"] > 0 else 0.0)}
        if replay is not None:
            n5 = min(args.steps, 5)
            # the replayed trajectory against the eager one of the headline run, step for step (same seeds, same data):
            # bit-equal since the control words are zeroed by a kernel (round 3: memset nodes)
            nr = len(replay["losses"])
            eager_same = m["losses"][:nr] if len(m["losses"]) >= nr else None
            out["hipgraph_replay"] = {"note": "same workload, the whole step (forward, loss, backward, Adam) captured once with "
                                              "GraphedTrainStep and replayed; ~550 launches per step lose ~20 us each on the eager path",
                                      "value": replay["voxels_per_step"] * n5 / replay["elapsed"], "unit": "HR voxels/s",
                                      "steps": n5, "warmup": 2, "ms_per_step": replay["elapsed"] / n5 * 1e3, "loss": replay["loss"],
                                      "losses": replay["losses"], "eager_losses_same_steps": eager_same,
                                      "loss_matches_eager": (eager_same == replay["losses"]) if eager_same is not None else None}
        if bf16_same is not None:
            n5 = min(args.steps, 5)
            bp = bf16_same["prof"]["hconv"]
            balgo = bp["work"] / (bp["ms"] * 1e-3) / 1e12 if bp["ms"] > 0 else 0.0
            bb = bf16_same["breakdown"]
            out["bf16_storage"] = {
                "workload": workload_name(args.lr_grid, bf16_same["hr"], batch, loss_name, world, "bf16"),
                "note": ("engine extension `model: {storage_dtype: bf16}`: same model, same shape as the headline; NOT the headline "
                         "(the reference and configs[1] are fp32).  Parity unpinned: the reference has no bf16 path; "
                         "tests/test_gpu_bf16.py holds every kernel to exactness on bf16-representable operands and the whole "
                         "model to 1e-2 (prediction, loss) / 5e-2 (parameter gradients, the bf16 run's branch decisions forced "
                         "into the oracle) of the fp32 oracle"),
                "value": bf16_same["voxels_per_step"] * n5 / bf16_same["elapsed"], "unit": "HR voxels/s", "steps": n5, "warmup": 1,
                "ms_per_step": bf16_same["elapsed"] / n5 * 1e3, "loss": bf16_same["loss"], "peak_mem_gb": bf16_same["peak_mem_gb"],
                "roofline": {"bound": "mfma", "kernel": "hconv_kernel<bf16>: one v_mfma_f32_16x16x32_bf16 per product group "
                                                        "(executed = algorithmic FLOPs; channel padding to 16 not counted)",
                             "achieved": balgo, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": balgo / F16_MFMA_PEAK_TFLOPS,
                             "frac_algorithmic": balgo / F16_MFMA_PEAK_TFLOPS, "executed_per_algorithmic": 1.0,
                             "kernel_ms_per_step": bp["ms"] / n5, "launches_per_step": bp["launches"] / n5},
                "conv_kernels": {k: {"ms_per_step": bb[k]["ms"] / bf16_same["breakdown_steps"],
                                     "algorithmic_tflops": (bb[k]["work"] / (bb[k]["ms"] * 1e-3) / 1e12 if bb[k]["ms"] > 0 else 0.0)}
                                 for k in ("hconv", "hconv_small", "igemm_s1", "igemm_s2", "igemm_bwd_s2", "wgrad")},
                "hbm_bound_kernels": {k: {"ms_per_step": bb[k]["ms"] / bf16_same["breakdown_steps"],
                                          "gbytes_per_s": (bb[k]["work"] / (bb[k]["ms"] * 1e-3) / 1e9 if bb[k]["ms"] > 0 else 0.0)}
                                      for k in ("loss", "act_bwd", "bias_grad", "adam", "data", "pack_reduce")}}
        if leg_errors:
            out["attached_leg_errors"] = leg_errors
        if bf16_c4 is not None:
            n3 = min(args.steps, 3)
            out["config4_bf16_hipgraph"] = {
                "workload": workload_name([40, 160, 160], bf16_c4["hr"], 1, "mixed", world, "bf16") + " [hipGraph replay]",
                "value": bf16_c4["voxels_per_step"] * n3 / bf16_c4["elapsed"], "unit": "HR voxels/s", "steps": n3, "warmup": 1,
                "ms_per_step": bf16_c4["elapsed"] / n3 * 1e3, "loss": bf16_c4["loss"], "peak_mem_gb": bf16_c4["peak_mem_gb"],
                "step_algorithmic_tflops": FLOP_PER_VOXEL * bf16_c4["voxels_per_step"] * n3 / bf16_c4["elapsed"] / 1e12}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m["cfg"])
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

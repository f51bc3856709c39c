#!/usr/bin/env python3
"""Experiment driver with the reference's surface (pytorch/script/train_model.py): same CLI flags
(--config_path, --world_size), same YAML, same outputs (<result_dir>/weights.pth = best state_dict,
learning_history.csv, log.txt), one process per GPU, DDP-style gradient averaging over RCCL.

Differences that make it runnable on ROCm and outside the authors' cluster:
  * no hard CUDA/NCCL/MLflow requirement: MLflow is used only if importable; the rendez-vous is 127.0.0.1;
  * --data_root / --result_root instead of paths derived from $PYTHONPATH (train_model.py:55-58);
  * the model runs on the sr3d HIP engine, DistributedDataParallel is replaced by GradAllReducer.

    python 3d-sr-micrometeorology_amd/script/train_model.py --config_path cfg.yml --world_size 1 \
           --data_root data/DL_data --result_root results
"""
import argparse
import copy
import logging
import os
import pathlib
import socket
import sys
import time
import traceback

ROOT = str(pathlib.Path(__file__).resolve().parents[2])
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402
import yaml  # noqa: E402

logger = logging.getLogger()


class _Both:
    """model optimizer (FlatAdam) + the GradNorm task weights' own Adam (train_model.py:185-199)"""

    def __init__(self, flat, extra):
        self.flat, self.extra = flat, extra

    def zero_grad(self):
        self.flat.zero_grad()
        self.extra.zero_grad(set_to_none=False)

    def step(self):
        self.flat.step()
        self.extra.step()

    @property
    def grad_scale(self):
        return self.flat.grad_scale

    @grad_scale.setter
    def grad_scale(self, v):
        self.flat.grad_scale = v


def train_and_validate(rank: int, world_size: int, config: dict, weight_path: str, learning_history_path: str,
                       data_root: str, port):
    import sr3d_amd
    from sr3d_amd.src.dataloader import data_dirs_of_config, make_dataloaders, split_into_train_valid_test_dirs
    from sr3d_amd.src.gradnorm import GradNorm
    from sr3d_amd.src.optim_helper import test_ddp, train_ddp
    from sr3d_amd.src.utils import set_seeds

    # spawned workers start with an unconfigured root logger: rank 0 appends to the run's log.txt
    if rank == 0:
        logging.basicConfig(level=logging.INFO, handlers=[
            logging.StreamHandler(sys.stdout), logging.FileHandler(os.path.join(os.path.dirname(weight_path), "log.txt"))])
    # rendezvous: a file next to the results (no TCP port to race for between `_free_port()` and the workers' bind: an
    # "address already in use" was seen once in a test run); an int is still taken as a TCP port on 127.0.0.1
    if isinstance(port, int):
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        init_method = None
    else:
        # absolute: "file://out/x" would parse as netloc "out" + path "/x" (a relative --result_root), and torch only uses the path
        init_method = "file://" + os.path.abspath(str(port))
    # RCCL ("nccl" under PyTorch-ROCm), one GPU per rank.  SR3D_DIST_BACKEND=gloo keeps the ranks on the CPU: the
    # multi-process rehearsal of this function's control flow (tests/test_dist_paths_gloo.py, with a stub engine -- the
    # HIP engine itself has no CPU path and raises).
    backend = os.environ.get("SR3D_DIST_BACKEND", "nccl")
    if backend == "nccl":
        device = torch.device("cuda", rank)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", init_method=init_method, rank=rank, world_size=world_size, device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group(backend, init_method=init_method, rank=rank, world_size=world_size)
    set_seeds(config["train"]["seed"])
    use_grad_norm = "grad_norm" in config["train"]

    all_data_dirs = data_dirs_of_config(config, pathlib.Path(data_root))
    split = split_into_train_valid_test_dirs(all_data_dirs, config["data"]["train_valid_test_ratios"])
    dataloaders, samplers = make_dataloaders(
        rank=rank, world_size=world_size, data_dirs=split, hr_3d_build_path=all_data_dirs[0].parent / "hr_is_in_build.npy",
        hr_org_size=tuple(config["data"]["hr_org_size"]), hr_crop_size=tuple(config["data"]["hr_crop_size"]),
        batch_size=config["data"]["batch_size"], means=config["data"]["means"], stds=config["data"]["stds"],
        nan_value=config["data"]["nan_value"], num_workers=config["data"].get("num_workers", 2),
        datasizes=config["data"]["datasizes"], seed=config["data"]["seed"], lr_scaling=config["data"].get("lr_scaling"),
        max_discarded_lr_z_index=config["data"].get("max_discarded_lr_z_index"),
        scale_factor=config["data"].get("scale_factor", 4),
        # engine extension (absent from the reference's YAML = off): normalise / clamp / NaN-fill on the GPU, one batch
        # ahead of the step (src/device_pipeline.py); the batches are bit-identical to the CPU pipeline's
        device_pipeline=device if config["data"].get("device_pipeline", False) else None)

    model = sr3d_amd.make_model(config).to(device)
    loss_fn = sr3d_amd.make_loss(config)
    # engine extension (absent from the reference's YAML = off): `train: {hip_graph: true}` replays the training step as a
    # hipGraph (src/graph.py; no GradNorm): 15 % faster on the reference's 32x64x64 crops.  With more than one rank the
    # gradient averaging is part of the replayed step (`hip_graph_comm: split | captured`, src/graph.py)
    use_graph = bool(config["train"].get("hip_graph", False)) and not use_grad_norm
    flat = sr3d_amd.FlatAdam(model.parameters(), lr=config["train"]["lr"], capturable=use_graph)
    graph_step = None
    reducer = sr3d_amd.GradAllReducer(flat.params, flat.flat_grad, flat.offsets)
    reducer.broadcast_parameters(flat.flat_param)   # what the DDP constructor does (train_model.py:179)
    if use_graph:
        from sr3d_amd.src.optim_helper import LazyGraphedStep
        graph_step = LazyGraphedStep(model, loss_fn, flat, reducer=reducer if world_size > 1 else None,
                                     comm=config["train"].get("hip_graph_comm"))
        if world_size == 1:
            reducer.remove_hooks()      # one rank: nothing to average, the captured step is the whole step
            reducer = None
    grad_norm, optimizer = None, flat
    if use_grad_norm:
        gn = config["train"]["grad_norm"]
        grad_norm = GradNorm(n_tasks=gn["n_tasks"], alpha=gn["alpha"], output_dir_path=os.path.dirname(weight_path),
                             device=device if device.type == "cpu" else rank, clipping_weight_min=gn.get("clipping_weight_min"))
        optimizer = _Both(flat, torch.optim.Adam([grad_norm.weights], lr=gn["lr"]))

    all_scores, best_loss = [], np.inf
    best_weights = copy.deepcopy(model.state_dict())
    for epoch in range(config["train"]["num_epochs"]):
        t0 = time.time()
        dist.barrier()
        loss = train_ddp(dataloader=dataloaders["train"], sampler=samplers["train"], model=model, loss_fn=loss_fn,
                         optimizer=optimizer, epoch=epoch, rank=device, world_size=world_size,
                         num_loops=config["train"]["num_loops_train"], grad_norm=grad_norm, reducer=reducer,
                         graph_step=graph_step)
        dist.barrier()
        val_loss = test_ddp(dataloader=dataloaders["valid"], sampler=samplers["valid"], model=model, loss_fn=loss_fn,
                            epoch=epoch, rank=device, world_size=world_size, num_loops=config["train"]["num_loops_valid"],
                            grad_norm=grad_norm)
        dist.barrier()
        all_scores.append({"loss": loss, "val_loss": val_loss})
        if use_grad_norm:
            grad_norm.record_and_write_out_weights_and_losses()
        if rank == 0:
            logger.info(f"Epoch: {epoch + 1}, loss = {loss:.8f}, val_loss = {val_loss:.8f}")
            if val_loss <= best_loss:
                best_loss, best_weights = val_loss, copy.deepcopy(model.state_dict())
                torch.save(best_weights, weight_path)
                logger.info("Best loss is updated.")
            if epoch % 10 == 0:
                pd.DataFrame(all_scores).to_csv(learning_history_path, index=False)
            logger.info(f"Elapsed time = {time.time() - t0} sec")
    if rank == 0:
        torch.save(best_weights, weight_path)
        pd.DataFrame(all_scores).to_csv(learning_history_path, index=False)
    dist.destroy_process_group()


def write_out_inferences(*, test_loader, model, inference_dir: str, device: str) -> None:
    """train_model.py:83-101: one forward pass per test sample (whole domain), saved next to its inputs as
    <time stamp>_{LR,BM,HR,SR}.npy in normalised units, with the sample's L1 error in the log"""
    import sr3d_amd
    l1 = sr3d_amd.make_loss({"train": {"loss": {"name": "L1"}}})
    os.makedirs(inference_dir, exist_ok=True)
    for hr_path, (Xs, bs, ys) in zip(test_loader.dataset.hr_files, test_loader):
        bs = bs.unsqueeze(1)
        with torch.no_grad():
            Xd, bd, yd = Xs.to(device), bs.to(device), ys.to(device)
            preds = model(Xd, bd)
            err = float(l1(preds, yd, bd))
        stamp = os.path.basename(hr_path).split("_")[0]
        for label, t in (("LR", Xs), ("BM", bs), ("HR", ys), ("SR", preds.cpu())):
            np.save(os.path.join(inference_dir, f"{stamp}_{label}.npy"), t.numpy())
        logger.info(f"{stamp}, l1 = {err:.7f}")


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_path", type=str, required=True)
    ap.add_argument("--world_size", type=int, default=2)
    ap.add_argument("--data_root", type=str, default=f"{ROOT}/data/DL_data")
    ap.add_argument("--result_root", type=str, default=f"{ROOT}/data/DL_results")
    ap.add_argument("--inference_root", type=str, default=f"{ROOT}/data/DL_inferences")
    args = ap.parse_args()

    with open(args.config_path) as f:
        config = yaml.safe_load(f)
    experiment_name = args.config_path.split("/")[-2] if "/" in args.config_path else "default"
    config_name = os.path.basename(args.config_path).split(".")[0]
    result_dir = f"{args.result_root}/{experiment_name}/{config_name}"
    os.makedirs(result_dir, exist_ok=False)     # the reference also refuses to overwrite (train_model.py:294)
    logging.basicConfig(level=logging.INFO, handlers=[logging.StreamHandler(sys.stdout),
                                                      logging.FileHandler(f"{result_dir}/log.txt")])
    weight_path, history_path = f"{result_dir}/weights.pth", f"{result_dir}/learning_history.csv"
    if not torch.cuda.is_available():
        raise Exception("No GPU.")
    assert args.world_size <= torch.cuda.device_count()
    try:
        import mlflow  # optional
        mlflow.set_tracking_uri(f"{args.result_root}/mlruns")
        mlflow.set_experiment(experiment_name)
        mlflow.start_run(run_name=config_name)
    except ImportError:
        mlflow = None
    try:
        t0 = time.time()
        rendezvous = os.path.abspath(os.path.join(result_dir, ".rendezvous"))
        if os.path.exists(rendezvous):
            os.remove(rendezvous)
        mp.spawn(train_and_validate, args=(args.world_size, config, weight_path, history_path, args.data_root,
                                           rendezvous), nprocs=args.world_size, join=True)
        if os.path.exists(rendezvous):
            os.remove(rendezvous)
        logger.info(f"Total elapsed time = {time.time() - t0} sec")

        # final evaluation on the test split (train_model.py:351-390): whole domain, the reference's ten metrics
        import sr3d_amd
        from sr3d_amd.src import loss_maker as lm
        from sr3d_amd.src.dataloader import make_evaluation_dataloader_without_random_cropping
        from sr3d_amd.src.optim_helper import evaluate
        model = sr3d_amd.make_model(config).to("cuda:0")
        model.load_state_dict(torch.load(weight_path, map_location="cuda:0"))
        model.eval()
        test_loader = make_evaluation_dataloader_without_random_cropping(config, pathlib.Path(args.data_root),
                                                                         batch_size=1, num_workers=0)
        stds = config["data"]["stds"]
        loss_fns = {
            "L1": lm.MyL1Loss(),
            "MaskedL1": lm.MaskedL1Loss(),
            "MaskedL1NearWall": lm.MaskedL1LossNearWall(),
            "ResidualContinuityEq": lm.ResidualContinuity(stds[1:]),
            "AbsDiffTemperature": lm.AbsDiffTemperature(stds[0]),
            "DiffVelocityNorm": lm.DiffVelocityVectorNorm(stds[1:]),
            "AbsDiffTemperatureLevZero": lm.AbsDiffTemperature(stds[0], lev=0),
            "DiffVelocityNormLevZero": lm.DiffVelocityVectorNorm(stds[1:], lev=0),
            "AbsDiffDivergence": lm.AbsDiffDivergence(stds[1:]),
            "DiffOmegaVectorNorm": lm.DiffOmegaVectorNorm(stds[1:]),
        }
        results = evaluate(dataloader=test_loader, model=model, loss_fns=loss_fns, device="cuda:0")
        if config["train"].get("write_out_inferences", False):        # train_model.py:391-397
            write_out_inferences(test_loader=test_loader, model=model, device="cuda:0",
                                 inference_dir=f"{args.inference_root}/{experiment_name}/{config_name}")
        for k, v in results.items():
            logger.info(f"{k}: {v.avg:.8f}")
            if mlflow is not None:
                mlflow.log_metric(k, v.avg)
    except Exception:
        logger.info("\n*********************************************************")
        logger.info("Error")
        logger.info("*********************************************************\n")
        logger.error(traceback.format_exc())
        raise
    finally:
        if mlflow is not None:
            mlflow.end_run()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Training entrypoint with the reference's command line (run_train.py of JinPLu/VeloxSeg), on the MI355X engine.

    python run_train.py --dataset_name AutoPETII --model_name VeloxSeg --train_config cfg/train.json --model_config cfg/models.json --synthetic 8

The reference's data side (MONAI NIfTI datasets + transforms, utils/train_*.py:97-212) is out of scope of this repository (SURVEY.md 2
rows 13-15): batches are synthetic patches (--synthetic STEPS_PER_EPOCH, default 8) unless a Python module providing
`build_loaders(args, train_config, model_config) -> (train_loader, val_loader)` is named with --data_module.  Everything after the
loader -- model, Loss, AdamW, schedulers, step loop, metrics, checkpoints -- follows utils/train_brats2021.py (veloxseg_amd/utils/train_loop.py).
For N GPUs: python -m torch.distributed.run --nproc-per-node N run_train.py ...   (one process per GPU, RCCL gradient all-reduce).
"""
import argparse
import importlib
import json
import logging
import os
import time

SUPPORTED_DATASETS = ("AutoPETII", "Hecktor2022", "BraTS2021")

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--dataset_name", type=str, required=True, choices=SUPPORTED_DATASETS, help="dataset name")
    parser.add_argument("--model_name", type=str, required=True, help="model name")
    parser.add_argument("--train_config", type=str, required=True, help="train_config path")
    parser.add_argument("--model_config", type=str, required=True, help="model_config path")
    parser.add_argument("--checkpoint_path", type=str, default=None, help="checkpoint path")
    parser.add_argument("--gpu_id", type=str, default="0", help="gpu id")
    parser.add_argument("--num_workers", type=int, default=8, help="number of workers for data loading")
    parser.add_argument("--model_index", type=str, default=None, help="Markdown index of the model")
    parser.add_argument("--select_modal", type=int, default=None)
    parser.add_argument("--synthetic", type=int, default=8, dest="synthetic_steps", help="synthetic batches per epoch")
    parser.add_argument("--augment", action="store_true", help="synthetic whole cases through the GPU transform chain (crop foreground, pos/neg crop, z-rotation) instead of ready-made patches")
    parser.add_argument("--data_module", type=str, default=None, help="python module with build_loaders(args, train_config, model_config)")
    parser.add_argument("--save_path", type=str, default=None, help="checkpoint directory (default ./checkpoints/<date>_<dataset>)")
    parser.add_argument("--graph", action="store_true", dest="use_graph", help="capture the training step once and replay it as a launch tape (csrc/tape.hip: ~2 ms of host time per step instead of ~10 ms; falls back to eager launches if the capture cannot be verified)")
    parser.add_argument("--precision", default="fp32", choices=["fp32", "bf16"], help="bf16 = the mode of BASELINE configs[1] (reference speed_test.py:122,127 autocast): 16-bit storage of the full-resolution heads and the 32^3-level JLC internals, bf16 matrix-pipe operands; fp32 statistics, loss, master weights (INTEGRATION.md)")
    args = parser.parse_args()
    with open(args.train_config, "r", encoding="utf-8") as f:
        train_config = json.load(f)
    with open(args.model_config, "r", encoding="utf-8") as f:
        model_config = json.load(f)
    import veloxseg_amd  # noqa: F401
    import torch
    import torch.distributed as dist
    from veloxseg_amd.utils.train_loop import run_train
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    local_rank = int(os.environ.get("LOCAL_RANK", args.gpu_id.split(",")[0]))
    torch.cuda.set_device(local_rank)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        dist.init_process_group("nccl")          # RCCL
    torch.manual_seed(12345)                     # utils/seed.py:6
    train_loader = val_loader = None
    if args.data_module:
        train_loader, val_loader = importlib.import_module(args.data_module).build_loaders(args, train_config, model_config)
    save = args.save_path or os.path.join("checkpoints", time.strftime("%Y-%m-%d_%H-%M-%S") + "_" + args.dataset_name)
    hist = run_train(args, train_config, model_config, train_loader, val_loader, save_path=save if int(os.environ.get("RANK", "0")) == 0 else None)
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"epochs": len(hist["loss"]), "loss": hist["loss"], "lr": hist["lr"], "best_train_dice": hist["best_train_dice"], "save_path": save}))
    if dist.is_initialized():
        dist.destroy_process_group()

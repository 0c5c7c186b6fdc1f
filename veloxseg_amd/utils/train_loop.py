"""Step-loop harness with the reference's training semantics (utils/train_brats2021.py:40-97,219-330 and the
near-identical train_autopet.py / train_hecktor.py), on TrainEngine.

Kept: model / Loss / optimizer / warm-up + training scheduler construction from the same config keys, resume from a reference-format
checkpoint, per step  zero_grad -> model -> Loss -> backward -> AdamW (one TrainEngine.step), per-step metrics of the segmentation heads
(output[:-(2+M)]), per-epoch scheduler stepping (warm-up LambdaLR then cosine / poly / plateau), `save_model_interval` / best-train /
best-val checkpoints in the reference format, validation every `val_interval` epochs in eval mode.
Not kept (out of scope): the MONAI NIfTI datasets / transforms, TensorBoard, the log-file layout.  Data comes from
any iterable of (inputs, labels) batches; `SyntheticPatches` is the stand-in used by run_train.py --synthetic and the tests.
"""
from __future__ import annotations

import logging
import os
import time
import types
from typing import Callable, Iterable, Optional

import torch

from ..engine import TrainEngine
from .load_model import load_checkpoint, load_model, save_checkpoint
from .loss import Loss
from .optimizers.optimizers import build_optimizer
from .optimizers.schedulers import build_scheduler, select_scheduler, step_scheduler

log = logging.getLogger("veloxseg_amd.train")


class SyntheticPatches:
    """`steps` batches of seeded random patches per epoch: x ~ N(0,1), labels = rand > 0.97 (binary) or randint(ncls)"""

    def __init__(self, model_cfg, batch, steps, device, seed=12345):
        self.cfg, self.batch, self.steps, self.device, self.seed = model_cfg, batch, steps, device, seed

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        S = self.cfg["input_size"]
        for _ in range(self.steps):
            x = torch.randn((self.batch, sum(self.cfg["in_ch"]), *S), generator=g)
            if self.cfg["n_classes"] == 2:
                y = (torch.rand((self.batch, 1, *S), generator=g) > 0.97).long()
            else:
                y = torch.randint(0, self.cfg["n_classes"], (self.batch, 1, *S), generator=g)
            yield x.to(self.device), y.to(self.device)


class SyntheticVolumes:
    """whole synthetic cases (padded background + a foreground body + lesion blobs) pushed through the reference's transform chain on the GPU
    (utils/augment.py: CropForegroundd -> RandCropByPosNegLabeld(num_samples) -> RandRotated): `batch` patches per step, batch / num_samples cases per step"""

    def __init__(self, model_cfg, batch, steps, device, volume=(160, 160, 144), num_samples=2, seed=12345, rotate_degrees=15.0):
        from . import augment as A
        if batch % num_samples:
            raise ValueError("batch must be a multiple of num_samples")
        self.cfg, self.batch, self.steps, self.device, self.volume, self.seed, self.ns = model_cfg, batch, steps, device, tuple(volume), seed, num_samples
        self.M = sum(model_cfg["in_ch"])
        self.keys = [f"img{m}" for m in range(self.M)] + ["seg"]
        self.pipe = A.Compose([
            A.CropForegroundd(self.keys, "img0"),
            A.RandCropByPosNegLabeld(self.keys, "seg", model_cfg["input_size"], pos=1, neg=1, num_samples=num_samples),
            A.RandRotated(self.keys, range_z=A.rotation_range_from_degrees(rotate_degrees), mode=A.image_label_modes(self.M), prob=0.5),
        ]).set_random_state(seed)

    def __len__(self):
        return self.steps

    def case(self, g):
        D, Hh, W = self.volume
        pad = 8
        body = (slice(None), slice(pad, D - pad), slice(pad, Hh - pad), slice(pad, W - pad))
        data = {}
        for m in range(self.M):
            x = torch.full((1, D, Hh, W), -2.0, device=self.device)
            x[body] = torch.randn((1, D - 2 * pad, Hh - 2 * pad, W - 2 * pad), generator=g, device=self.device).abs() - 1.9
            data[f"img{m}"] = x
        seg = torch.zeros((1, D, Hh, W), device=self.device)
        ncls = self.cfg["n_classes"]
        for _ in range(6):
            c = [int(torch.randint(pad + 6, n - pad - 6, (1,), generator=g, device=self.device)) for n in (D, Hh, W)]
            cls = 1 if ncls == 2 else int(torch.randint(1, ncls, (1,), generator=g, device=self.device))
            seg[:, c[0] - 5:c[0] + 5, c[1] - 5:c[1] + 5, c[2] - 5:c[2] + 5] = float(cls)
        data["seg"] = seg
        return data

    def __iter__(self):
        g = torch.Generator(device=self.device).manual_seed(self.seed)
        for _ in range(self.steps):
            xs, ys = [], []
            for _ in range(self.batch // self.ns):
                for d in self.pipe(self.case(g)):
                    xs.append(torch.cat([d[k] for k in self.keys[:-1]], 0))
                    ys.append(d["seg"].long())
            yield torch.stack(xs), torch.stack(ys)


def _metric_fns(dataset_name):
    if dataset_name == "BraTS2021":
        from .metric.metrics_brats import show_deep_metrics
    else:
        from .metric.metrics import show_deep_metrics
    return show_deep_metrics


def run_train(args, train_config, model_config, train_loader: Optional[Iterable] = None, val_loader: Optional[Iterable] = None,
              save_path: Optional[str] = None, on_step: Optional[Callable] = None):
    """-> dict(epoch losses, learning rates, best dice, last checkpoint path).  `args` carries model_name, dataset_name, checkpoint_path
    (as run_train.py parses them); loaders yield (inputs, labels) CUDA tensors of a fixed shape."""
    device = torch.device("cuda", torch.cuda.current_device())
    model = load_model(args.model_name, model_config).to(device)
    mcfg = model_config[args.model_name]
    num_modal = len(mcfg["in_ch"])
    criterion = Loss(args, train_config, device, num_modal)
    optimizer = build_optimizer(model=model, optimizer_type=train_config["optimizer"]["optimizer_type"], optimizer_args=train_config["optimizer"]["optimizer_args"])
    warmup_scheduler = build_scheduler(optimizer=optimizer, scheduler_type="warmup_scheduler", config=train_config)
    training_scheduler = build_scheduler(optimizer=optimizer, scheduler_type="training_scheduler", config=train_config)
    warmup_epoch = train_config["warmup_scheduler"]["warmup_epochs"]
    start_epoch, best_train_dice, best_val_dice = 0, 0, 0
    if getattr(args, "checkpoint_path", None) is not None:
        model, optimizer, warmup_scheduler, training_scheduler, start_epoch, best_train_dice, best_val_dice = load_checkpoint(
            model, args.checkpoint_path, optimizer, warmup_scheduler, training_scheduler, device)
        log.info("Load Checkpoint, Continue to Train!!!!")
    batch = train_config["batch_size"]
    import torch.distributed as dist
    ddp = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if ddp else 0
    if train_loader is None:
        # every rank draws its OWN synthetic stream (seed + rank): data parallelism must add data, not replicate the same batch on every GPU
        source = SyntheticVolumes if getattr(args, "augment", False) else SyntheticPatches
        train_loader = source(mcfg, batch, getattr(args, "synthetic_steps", 4), device, seed=12345 + rank)
    # engine buffers from the configuration, not from a pulled batch (that would consume one batch of a one-shot iterable and spin up an extra
    # worker pool); loaders must yield (batch_size, sum(in_ch), *input_size) inputs -- labels of any integer dtype are converted by the copy
    size = list(mcfg["input_size"])
    x_shape = (batch, sum(mcfg["in_ch"]), *size)
    # per-head metrics (show_deep_metric) need every head at full resolution in engine.last_outputs: then the up-sampling is not fused into the loss
    engine = TrainEngine(model, criterion, x_shape, label_dtype=None, optimizer=optimizer, use_graph=getattr(args, "use_graph", False),
                         fuse_ds=not train_config.get("show_deep_metric", False), precision=getattr(args, "precision", "fp32"),
                         pipeline_tail=getattr(args, "pipeline_tail", False))  # (opt-in: the decoder tail of step N beside the encoder forward of step N + 1; flush() below)
    show_deep_metrics = _metric_fns(args.dataset_name)
    not_pred = 2 + num_modal if args.model_name == "VeloxSeg" else 0
    sched_type = train_config["train_scheduler"]["scheduler_type"]
    hist = {"loss": [], "lr": [], "dice": [], "val_dice": [], "checkpoints": []}
    if save_path:
        os.makedirs(save_path, exist_ok=True)
    for epoch in range(start_epoch, train_config["epochs"]):
        scheduler = select_scheduler(epoch, warmup_epoch, warmup_scheduler, training_scheduler)
        start = time.time()
        model.train()
        total_loss, total_dice, nsteps = 0.0, 0.0, 0
        hist["lr"].append(optimizer.param_groups[0]["lr"])
        for step, (inputs, labels) in enumerate(train_loader):
            loss = engine.step(inputs, labels)                    # zero_grad(set_to_none) -> fwd -> loss -> bwd -> AdamW (train_brats2021.py:232-239)
            l = loss.item()
            outs = engine.last_outputs
            metrics, string = show_deep_metrics(outs[:-not_pred] if not_pred else outs, labels, train_config["show_deep_metric"])
            log.info(f"train {epoch + 1}/{train_config['epochs']} {step}/{len(train_loader)} Training Loss:{l:.4f}\n" + string)
            total_loss += l
            total_dice += metrics[0] if args.dataset_name == "BraTS2021" else metrics[3]
            nsteps += 1
            if on_step is not None:
                on_step(epoch, step, l, metrics)
        engine.flush()            # the decoder half of the last update: before checkpoints / validation read the parameters on this stream
        if epoch < warmup_epoch:
            step_scheduler(scheduler, "warmup_scheduler")
        elif sched_type != "reducelronplateau":
            step_scheduler(scheduler, sched_type)
        mean_loss, mean_dice = total_loss / max(nsteps, 1), total_dice / max(nsteps, 1)
        hist["loss"].append(mean_loss)
        hist["dice"].append(mean_dice)
        if save_path and epoch % train_config["save_model_interval"] == 0:
            f = os.path.join(save_path, f"{epoch}.pth")
            save_checkpoint(model, optimizer, warmup_scheduler, training_scheduler, epoch, best_train_dice, best_val_dice, f)
            hist["checkpoints"].append(f)
        if mean_dice >= best_train_dice:
            best_train_dice = mean_dice
            if save_path:
                save_checkpoint(model, optimizer, warmup_scheduler, training_scheduler, epoch, best_train_dice, best_val_dice, os.path.join(save_path, "train_best.pth"))
        log.info(f"training epoch {epoch + 1}: loss {mean_loss:.4f} dice {mean_dice:.4f} best {best_train_dice:.4f} time {time.time() - start:.2f} s")
        if val_loader is not None and (epoch + 1) % train_config["val_interval"] == 0:
            model.eval()
            tot, n = 0.0, 0
            with torch.no_grad():
                for inputs, labels in val_loader:
                    out = model(inputs)
                    metrics, _ = show_deep_metrics(out, labels, False)
                    tot += metrics[0] if args.dataset_name == "BraTS2021" else metrics[3]
                    n += 1
            val_dice = tot / max(n, 1)
            if ddp:      # the LR plateau scheduler and the best-checkpoint decision must see the same number on every rank
                t_ = torch.tensor([tot, float(n)], device=device, dtype=torch.float64)
                dist.all_reduce(t_)
                val_dice = float(t_[0] / max(float(t_[1]), 1.0))
            hist["val_dice"].append(val_dice)
            if sched_type == "reducelronplateau" and epoch >= warmup_epoch:
                step_scheduler(training_scheduler, sched_type, val_dice)
            if val_dice >= best_val_dice:
                best_val_dice = val_dice
                if save_path:
                    save_checkpoint(model, optimizer, warmup_scheduler, training_scheduler, epoch, best_train_dice, best_val_dice, os.path.join(save_path, "val_best.pth"))
    hist.update(best_train_dice=best_train_dice, best_val_dice=best_val_dice, engine=engine)
    return hist

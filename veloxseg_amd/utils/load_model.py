"""Model registry / checkpoint helpers (reference: utils/load_model.py:83-148), VeloxSeg entry only."""
from collections import OrderedDict

import torch


def load_model(model_name, config):
    if model_name != "VeloxSeg":
        raise ValueError("Invalid model name, now {} (veloxseg_amd provides VeloxSeg only)".format(model_name))
    from ..model.VeloxSeg import VeloxSeg
    return VeloxSeg(**config[model_name])


def checkpoint_DDP_to_SingleGPU(checkpoint):
    out = OrderedDict()
    for k, v in checkpoint.items():
        out[k[len("module."):] if k.startswith("module.") else k] = v
    return out


def save_checkpoint(model, optimizer, warmup_scheduler, training_scheduler, epoch, best_train_dice, best_val_dice, filename="checkpoint.pth"):
    torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(), "warmup_scheduler": warmup_scheduler.state_dict(),
                "training_scheduler": training_scheduler.state_dict(), "epoch": epoch + 1, "best_train_dice": best_train_dice,
                "best_val_dice": best_val_dice}, filename)


def load_checkpoint(model, filename, optimizer=None, warmup_scheduler=None, training_scheduler=None, device=torch.device("cpu")):
    ckpt = torch.load(filename, map_location=device)
    model.load_state_dict(checkpoint_DDP_to_SingleGPU(ckpt["model"]))
    if optimizer is None:
        return model
    optimizer.load_state_dict(ckpt["optimizer"])
    for state in optimizer.state.values():
        for k, v in state.items():
            if isinstance(v, torch.Tensor):
                state[k] = v.to(device)
    warmup_scheduler.load_state_dict(ckpt["warmup_scheduler"])
    training_scheduler.load_state_dict(ckpt["training_scheduler"])
    return model, optimizer, warmup_scheduler, training_scheduler, ckpt["epoch"], ckpt["best_train_dice"], ckpt["best_val_dice"]

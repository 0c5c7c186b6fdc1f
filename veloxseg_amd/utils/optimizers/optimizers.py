"""Optimizer registry -- mirrors utils/optimizers/optimizers.py of the reference (same names, same torch.optim objects).
With TrainEngine the AdamW object is the STATE CARRIER (param_groups drive the LR schedulers, .state aliases the engine's flat
moment buffers so optimizer.state_dict() is the reference checkpoint format); the update itself is the fused HIP vx_adamw_step."""
from typing import Dict

import torch.optim as optim


def optim_adam(model, optimizer_args):
    return optim.Adam(model.parameters(), lr=optimizer_args["lr"], weight_decay=optimizer_args.get("weight_decay"))


def optim_sgd(model, optimizer_args):
    return optim.SGD(model.parameters(), lr=optimizer_args["lr"], weight_decay=optimizer_args.get("weight_decay"), momentum=optimizer_args.get("momentum"))


def optim_adamw(model, optimizer_args):
    return optim.AdamW(model.parameters(), lr=optimizer_args["lr"], weight_decay=optimizer_args["weight_decay"])


def build_optimizer(model, optimizer_type: str, optimizer_args: Dict):
    if optimizer_type == "adam":
        return optim_adam(model, optimizer_args)
    elif optimizer_type == "adamw":
        return optim_adamw(model, optimizer_args)
    elif optimizer_type == "sgd":
        return optim_sgd(model, optimizer_args)
    raise ValueError("must be adam or adamw for now")

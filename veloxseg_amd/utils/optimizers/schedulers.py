"""LR schedulers -- mirrors utils/optimizers/schedulers.py (names, config keys, per-epoch stepping).  The reference passes `verbose=False`
to LambdaLR / ReduceLROnPlateau (schedulers.py:19,35), a keyword torch >= 2.7 no longer accepts; it is dropped here."""
import torch.optim as optim


def warmup_lr_scheduler(config, optimizer):
    """lr = base * (epoch + 1) / warmup_epochs  (schedulers.py:16-20)"""
    return optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: (epoch + 1) * 1.0 / config["warmup_scheduler"]["warmup_epochs"])


def training_lr_scheduler(config, optimizer):
    scheduler_type = config["train_scheduler"]["scheduler_type"]
    if scheduler_type == "reducelronplateau":
        return optim.lr_scheduler.ReduceLROnPlateau(optimizer, factor=0.1, mode=config["train_scheduler"]["mode"],
                                                    patience=config["train_scheduler"]["patience"],
                                                    min_lr=config["train_scheduler"]["scheduler_args"]["min_lr"])
    elif scheduler_type == "cosine_annealing":
        return optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=config["train_scheduler"]["scheduler_args"]["epochs"],
                                                    eta_min=config["train_scheduler"]["scheduler_args"]["min_lr"])
    elif scheduler_type == "poly_lr":
        return optim.lr_scheduler.PolynomialLR(optimizer=optimizer, total_iters=config["epochs"] - config["warmup_scheduler"]["warmup_epochs"],
                                               power=config["train_scheduler"]["scheduler_args"]["power"], last_epoch=-1)
    raise NotImplementedError("Specified Scheduler Is Not Implemented")


def build_scheduler(optimizer, scheduler_type: str, config):
    if scheduler_type == "warmup_scheduler":
        return warmup_lr_scheduler(config=config, optimizer=optimizer)
    elif scheduler_type == "training_scheduler":
        return training_lr_scheduler(config=config, optimizer=optimizer)
    raise ValueError("Invalid Input -- Check scheduler_type")


def select_scheduler(epoch, warmup_epoch, warmup_scheduler, training_scheduler):
    return warmup_scheduler if epoch < warmup_epoch else training_scheduler


def step_scheduler(scheduler, scheduler_type, validation_metric=None):
    if scheduler_type == "reducelronplateau":
        scheduler.step(validation_metric)
    else:
        scheduler.step()

"""CPU: host logic of the step-loop harness: scheduler sequences, AdamW state carried in the reference's checkpoint
format, config-driven construction.  No kernels are launched."""
import math
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from recipe import CASES  # noqa: E402

TRAIN_CFG = {"batch_size": 2, "epochs": 8, "deep_Loss_weight": [1, 1, 1, 1], "RC_Loss_weight": 0.5, "Feature_Loss_weight": 2.0, "show_deep_metric": True,
             "save_model_interval": 1, "val_interval": 2, "optimizer": {"optimizer_type": "adamw", "optimizer_args": {"lr": 2.5e-4, "weight_decay": 0.01}},
             "warmup_scheduler": {"enabled": True, "warmup_epochs": 3}, "train_scheduler": {"scheduler_type": "cosine_annealing", "scheduler_args": {"epochs": 5, "min_lr": 1e-6}}}


def test_warmup_then_cosine_lr_sequence_follows_the_reference_loop():
    """LambdaLR((e+1)/W) for the first W epochs, then CosineAnnealingLR, each stepped once per epoch (train_brats2021.py:222,262-265)"""
    from veloxseg_amd.utils.optimizers.optimizers import build_optimizer
    from veloxseg_amd.utils.optimizers.schedulers import build_scheduler, select_scheduler, step_scheduler
    model = torch.nn.Linear(3, 2)
    opt = build_optimizer(model, "adamw", TRAIN_CFG["optimizer"]["optimizer_args"])
    assert isinstance(opt, torch.optim.AdamW) and opt.param_groups[0]["weight_decay"] == 0.01
    warm = build_scheduler(opt, "warmup_scheduler", TRAIN_CFG)
    train = build_scheduler(opt, "training_scheduler", TRAIN_CFG)
    W, base = 3, 2.5e-4
    lrs = []
    for epoch in range(8):
        sch = select_scheduler(epoch, W, warm, train)
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        step_scheduler(sch, "warmup_scheduler" if epoch < W else "cosine_annealing")
    # warm-up epochs see base*(e+1)/W; the cosine schedule then continues from the last warm-up value with T_max 5, eta_min 1e-6
    assert all(abs(lrs[e] - base * (e + 1) / W) < 1e-12 for e in range(W))
    assert abs(lrs[W] - base * (W + 1) / W) < 1e-12          # LambdaLR was stepped W times: factor (W+1)/W is what the reference hands over
    assert all(lrs[e + 1] < lrs[e] for e in range(W, 7)), lrs
    assert lrs[-1] > 1e-6


def test_engine_moments_travel_in_the_reference_checkpoint_format(tmp_path):
    from veloxseg_amd.engine import TrainEngine
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    from veloxseg_amd.utils.load_model import load_checkpoint, save_checkpoint
    from veloxseg_amd.utils.optimizers.optimizers import build_optimizer
    from veloxseg_amd.utils.optimizers.schedulers import build_scheduler
    cfg, _ = CASES["g2_32_m2"]
    torch.manual_seed(0)
    model = VeloxSeg(**cfg)
    opt = build_optimizer(model, "adamw", TRAIN_CFG["optimizer"]["optimizer_args"])
    warm, train = build_scheduler(opt, "warmup_scheduler", TRAIN_CFG), build_scheduler(opt, "training_scheduler", TRAIN_CFG)
    eng = TrainEngine(model, None, (1, 2, 32, 32, 32), optimizer=opt)
    eng.m.copy_(torch.randn_like(eng.m))
    eng.v.copy_(torch.rand_like(eng.v))
    eng.t = 7
    eng._step_tensor.fill_(7.0)
    f = str(tmp_path / "ck.pth")
    save_checkpoint(model, opt, warm, train, 4, 0.5, 0.25, f)
    ck = torch.load(f)
    assert set(ck) == {"model", "optimizer", "warmup_scheduler", "training_scheduler", "epoch", "best_train_dice", "best_val_dice"} and ck["epoch"] == 5
    assert len(ck["optimizer"]["state"]) == len(list(model.parameters()))
    # a plain torch.optim.AdamW (what the reference builds) loads it ...
    torch.manual_seed(1)
    model2 = VeloxSeg(**cfg)
    opt2 = build_optimizer(model2, "adamw", TRAIN_CFG["optimizer"]["optimizer_args"])
    warm2, train2 = build_scheduler(opt2, "warmup_scheduler", TRAIN_CFG), build_scheduler(opt2, "training_scheduler", TRAIN_CFG)
    _, opt2, _, _, epoch, btd, bvd = load_checkpoint(model2, f, opt2, warm2, train2)
    assert (epoch, btd, bvd) == (5, 0.5, 0.25)
    for p1, p2 in zip(model.parameters(), model2.parameters()):
        assert torch.equal(p1, p2)
        assert torch.equal(opt.state[p1]["exp_avg"], opt2.state[p2]["exp_avg"]) and float(opt2.state[p2]["step"]) == 7.0
    # ... and an engine bound to the loaded optimizer picks the moments and the step count up
    eng2 = TrainEngine(model2, None, (1, 2, 32, 32, 32), optimizer=opt2)
    assert eng2.t == 7
    for n in eng.flat.names:                      # (the 64-element alignment gaps of the flat buffers carry no state)
        o, k = eng.flat.slices[n]
        assert torch.equal(eng2.m[o:o + k], eng.m[o:o + k]) and torch.equal(eng2.v[o:o + k], eng.v[o:o + k]), n
    # the carrier's hyper-parameters are what the fused update reads
    opt2.param_groups[0]["lr"] = 1e-3
    g = opt2.param_groups[0]
    assert (g["betas"], g["eps"]) == ((0.9, 0.999), 1e-8)


def test_synthetic_patches_are_seeded_and_shaped():
    from veloxseg_amd.utils.train_loop import SyntheticPatches
    cfg, _ = CASES["g2_32_m2"]
    a = list(SyntheticPatches(cfg, 2, 3, "cpu"))
    b = list(SyntheticPatches(cfg, 2, 3, "cpu"))
    assert len(a) == 3 and a[0][0].shape == (2, 2, 32, 32, 32) and a[0][1].shape == (2, 1, 32, 32, 32) and a[0][1].dtype == torch.int64
    assert all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(a, b))
    assert not torch.equal(a[0][0], a[1][0])

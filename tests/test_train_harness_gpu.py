"""GPU: the step-loop harness end to end on synthetic patches -- epochs, metrics, reference-format checkpoints, resume -- and the fused
HIP AdamW driven through a torch.optim.AdamW state carrier against torch's own AdamW.step() on the same gradients."""
import copy
import json
import os
import subprocess
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
from recipe import CASES  # noqa: E402
from test_train_harness_cpu import TRAIN_CFG  # noqa: E402


def _args(**kw):
    return types.SimpleNamespace(model_name="VeloxSeg", dataset_name=kw.pop("dataset_name", "AutoPETII"), checkpoint_path=None, synthetic_steps=3, **kw)


def test_run_train_synthetic_checkpoints_and_resume(tmp_path):
    from veloxseg_amd.utils.train_loop import run_train
    cfg, _ = CASES["g2_32_m2"]
    mc = {"VeloxSeg": cfg}
    tc = dict(TRAIN_CFG, epochs=3)
    torch.manual_seed(12345)
    h = run_train(_args(), tc, mc, save_path=str(tmp_path / "run"))
    assert len(h["loss"]) == 3 and h["loss"][-1] < h["loss"][0], h["loss"]           # same 3 batches every epoch: the loss must go down
    assert abs(h["lr"][0] - 2.5e-4 / 3) < 1e-12 and abs(h["lr"][1] - 2.5e-4 * 2 / 3) < 1e-12
    assert all(os.path.exists(os.path.join(str(tmp_path / "run"), f)) for f in ("0.pth", "1.pth", "2.pth", "train_best.pth"))
    ck = torch.load(os.path.join(str(tmp_path / "run"), "1.pth"))
    assert ck["epoch"] == 2 and float(ck["optimizer"]["state"][0]["step"]) == 6.0                # 2 epochs x 3 steps
    # resume from the epoch-1 checkpoint: epoch 2 is replayed with the same data -> same loss as the uninterrupted run
    a = _args()
    a.checkpoint_path = os.path.join(str(tmp_path / "run"), "1.pth")
    h2 = run_train(a, tc, mc, save_path=str(tmp_path / "resumed"))
    assert len(h2["loss"]) == 1
    assert abs(h2["loss"][0] - h["loss"][2]) <= 2e-4 * abs(h["loss"][2]), (h2["loss"], h["loss"])
    assert abs(h2["lr"][0] - h["lr"][2]) < 1e-12
    # BraTS metrics branch
    cfg4 = dict(cfg, in_ch=[2], n_classes=4)
    h3 = run_train(_args(dataset_name="BraTS2021"), dict(tc, epochs=1), {"VeloxSeg": cfg4})
    assert len(h3["dice"]) == 1 and 0.0 <= h3["dice"][0] <= 1.0


def test_run_train_with_launch_tapes_tracks_the_eager_harness(tmp_path):
    """run_train --graph: the step replayed as launch tapes (static buffers, LR from the scheduler through the state carrier, per-step metrics read
    from the engine's static outputs) gives the loss / Dice history of the eager harness"""
    from veloxseg_amd import functional as VF
    from veloxseg_amd.utils.train_loop import run_train
    cfg, _ = CASES["g2_32_m2"]
    tc = dict(TRAIN_CFG, epochs=2)
    hist = {}
    for tape in (False, True):
        VF.reset_dropout_sites()
        torch.manual_seed(12345)
        hist[tape] = run_train(_args(use_graph=tape), tc, {"VeloxSeg": cfg})
    for k in ("loss", "dice"):
        for a, b in zip(hist[True][k], hist[False][k]):
            assert abs(a - b) <= 2e-3 * max(abs(b), 1e-3), (k, hist[True][k], hist[False][k])
    assert hist[True]["lr"] == hist[False]["lr"]


def test_fused_adamw_through_the_carrier_equals_torch_adamw():
    from veloxseg_amd.engine import TrainEngine
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    from veloxseg_amd.utils.loss import Loss
    from veloxseg_amd.utils.optimizers.optimizers import build_optimizer
    from veloxseg_amd.utils.train_loop import SyntheticPatches
    cfg, _ = CASES["g2_32_m2"]
    torch.manual_seed(3)
    model = VeloxSeg(**cfg).cuda()
    ref_model = copy.deepcopy(model)
    crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), TRAIN_CFG, None, num_modal=2)
    opt = build_optimizer(model, "adamw", {"lr": 1e-3, "weight_decay": 0.05})
    ref_opt = build_optimizer(ref_model, "adamw", {"lr": 1e-3, "weight_decay": 0.05})
    eng = TrainEngine(model, crit, (2, 2, 32, 32, 32), optimizer=opt)
    ref_model.train()
    for step, (x, y) in enumerate(SyntheticPatches(cfg, 2, 3, "cuda")):
        if step == 2:
            opt.param_groups[0]["lr"] = ref_opt.param_groups[0]["lr"] = 3e-4          # a scheduler step in between
        eng.step(x, y)
        ref_opt.zero_grad(set_to_none=True)
        crit(ref_model(x), y, sr_labels=x).backward()
        ref_opt.step()
    torch.cuda.synchronize()
    checked = 0
    for (n, p), q in zip(model.named_parameters(), ref_model.parameters()):
        # biases in front of an InstanceNorm (and key biases) have a mathematically zero gradient: what reaches Adam is round-off noise and
        # its first steps are sign-like, so those elements are excluded (same rule as tests/test_dp_gpu.py)
        mask = q.grad.abs() > 1e-5 * max(1.0, float(q.grad.abs().max()))
        if bool(mask.any()) and float(q.grad.abs().max()) > 1e-4:
            checked += int(mask.sum())
            d_ = (p.detach() - q.detach())[mask].abs()
            # Adam normalises by sqrt(v): where |g| is small its update amplifies the summation-order noise of the gradient, so the bound is
            # a fraction of the total step length (3 steps x lr 1e-3) for the worst element and ~fp32 round-off on average
            assert float(d_.max()) <= 2e-4 and float(d_.mean()) <= 2e-6, (n, float(d_.max()), float(d_.mean()))
    assert checked > 1000
    st, rst = opt.state_dict()["state"], ref_opt.state_dict()["state"]
    assert float(st[0]["step"]) == float(rst[0]["step"]) == 3.0
    k = max(st)
    assert float((st[k]["exp_avg"].cuda() - rst[k]["exp_avg"]).abs().max()) <= 1e-6 + 1e-3 * float(rst[k]["exp_avg"].abs().max())


@pytest.mark.timeout(600)
def test_entrypoints_run_with_the_reference_command_line(tmp_path):
    cfg, _ = CASES["g2_32_m2"]
    mc, tc, te = tmp_path / "models.json", tmp_path / "train.json", tmp_path / "test.json"
    mc.write_text(json.dumps({"VeloxSeg": cfg}))
    tc.write_text(json.dumps(dict(TRAIN_CFG, epochs=2, patch_size={"AutoPETII": [32, 32, 32]})))
    te.write_text(json.dumps({"sliding_window": {"overlap": 0.25}}))
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_train.py"), "--dataset_name", "AutoPETII", "--model_name", "VeloxSeg", "--train_config", str(tc),
                        "--model_config", str(mc), "--synthetic", "2", "--save_path", str(tmp_path / "ck")], capture_output=True, text=True, env=env, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["epochs"] == 2 and os.path.exists(tmp_path / "ck" / "train_best.pth")
    # the same command with the step replayed as launch tapes and the bf16 opt-in mode (flags of this library, not of the reference)
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "run_train.py"), "--dataset_name", "AutoPETII", "--model_name", "VeloxSeg", "--train_config", str(tc),
                         "--model_config", str(mc), "--synthetic", "2", "--save_path", str(tmp_path / "ck2"), "--graph", "--precision", "bf16"],
                        capture_output=True, text=True, env=env, timeout=500)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert json.loads(r2.stdout.strip().splitlines()[-1])["epochs"] == 2 and "falling back" not in r2.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_test.py"), "--dataset_name", "AutoPETII", "--model_name", "VeloxSeg", "--train_config", str(tc),
                        "--model_config", str(mc), "--test_config", str(te), "--checkpoint_dir", str(tmp_path / "ck"), "--checkpoint_index", "train_best",
                        "--synthetic", "1", "--volume_shape", "48", "40", "36", "--out_csv", str(tmp_path / "res.csv")], capture_output=True, text=True, env=env, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.exists(tmp_path / "res.csv") and "synthetic_0" in r.stdout


def test_train_on_whole_cases_through_the_gpu_transform_chain(tmp_path):
    """--augment: synthetic whole cases -> crop foreground -> pos/neg label crops -> z-rotation, all on the GPU, feeding the step loop"""
    from veloxseg_amd.utils.train_loop import SyntheticVolumes, run_train
    cfg, _ = CASES["g2_32_m2"]
    src = SyntheticVolumes(cfg, 2, 2, torch.device("cuda", 0), volume=(72, 64, 56))
    batches = list(src)
    assert len(batches) == 2
    for x, y in batches:
        assert x.shape == (2, 2, 32, 32, 32) and y.shape == (2, 1, 32, 32, 32) and y.dtype == torch.long
        assert set(y.unique().tolist()) <= {0, 1}
    assert any(int(y.sum()) > 0 for _, y in batches)                  # pos/neg sampling found lesion voxels
    again = list(SyntheticVolumes(cfg, 2, 2, torch.device("cuda", 0), volume=(72, 64, 56)))
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(batches, again))      # seeded: reproducible
    h = run_train(_args(augment=True), dict(TRAIN_CFG, epochs=2), {"VeloxSeg": cfg}, save_path=str(tmp_path / "aug"))
    assert len(h["loss"]) == 2 and all(l == l for l in h["loss"])

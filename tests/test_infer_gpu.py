"""GPU: sliding-window inference and label metrics through the C ABI (csrc/infer.hip) against the oracle / the reference goldens."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sliding_window_oracle as SO  # noqa: E402  (checker only)

HERE = os.path.dirname(os.path.abspath(__file__))


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _pointwise(w):
    # a deterministic per-window function that also depends on the position INSIDE the window, so blending really averages different values
    ramp = torch.linspace(0, 1, w.shape[-1], device=w.device).view(1, 1, 1, 1, -1)
    return torch.cat([w[:, :1] * 2.0 + ramp, w[:, -1:] - 1.0, w.mean(1, keepdim=True) * ramp], 1)


@pytest.mark.parametrize("shape,roi,swb,overlap", [((2, 3, 20, 17, 9), (8, 8, 8), 3, 0.5), ((1, 2, 5, 6, 7), (8, 4, 16), 2, 0.25),
                                                   ((1, 4, 40, 33, 21), (16, 16, 16), 2, 0.5), ((3, 1, 12, 12, 12), (12, 12, 12), 4, 0.5)])
def test_sliding_window_matches_oracle_bit_exactly(shape, roi, swb, overlap):
    from veloxseg_amd.utils import inference_runtime as IR
    d = dev()
    x = torch.randn(shape, generator=torch.Generator().manual_seed(3)).to(d)
    want = SO.sliding_window_inference(x, roi, swb, _pointwise, overlap)          # oracle driver, same predictor, same device arithmetic
    got, labels = IR.sliding_window_inference(x, roi, swb, _pointwise, overlap=overlap, return_labels=True)
    assert got.shape == want.shape
    assert torch.equal(got, want), float((got - want).abs().max())
    assert labels.dtype == torch.uint8 and torch.equal(labels.long(), want.argmax(1, keepdim=True))
    cfgd = {"sliding_window": {"overlap": overlap}}
    assert torch.equal(IR.sliding_window_predict(x, _pointwise, roi, swb, cfgd), want)
    with pytest.raises(NotImplementedError):
        IR.sliding_window_inference(x, roi, swb, _pointwise, overlap=overlap, mode="gaussian")
    with pytest.raises(RuntimeError):
        IR.sliding_window_inference(x.cpu(), roi, swb, _pointwise, overlap=overlap)


def test_taped_predictor_keeps_weight_images_only_while_the_weights_stand(golden_dir):
    """engine.TapedPredictor builds the JLC blocks' weight images once and captures its tapes WITHOUT their preparation launches; the images must not outlive the weights:
    an in-place torch update (version counters) and a TrainEngine step (fused AdamW through raw pointers: the weights epoch) both make the next call re-capture.
    Every call is compared with the plain eager forward of the same weights."""
    import sys
    import types
    sys.path.insert(0, golden_dir)
    from recipe import CASES, fill_state_dict
    from veloxseg_amd.engine import TapedPredictor, TrainEngine
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    from veloxseg_amd.utils.loss import Loss
    cfg, _ = CASES["g2_32_m2"]
    d = dev()
    model = VeloxSeg(**cfg)
    sd = model.state_dict()
    fill_state_dict(sd)
    model.load_state_dict(sd)
    model = model.to(d).eval()
    x = torch.randn((2, 2, 32, 32, 32), generator=torch.Generator().manual_seed(7)).to(d)
    tp = TapedPredictor(model)

    def both():
        with torch.inference_mode():
            a = tp(x).float().clone()
            b = model(x)
            b = (b[0] if isinstance(b, (list, tuple)) else b).float().clone()
        return a, b
    a0, b0 = both()
    assert torch.equal(a0, b0)
    a0b, _ = both()                                            # a replay
    assert torch.equal(a0b, a0) and any(e is not None for e in tp._tapes.values())
    with torch.no_grad():                                      # in place: same storage, version counters move
        for p_ in model.parameters():
            p_.mul_(1.25)
    a1, b1 = both()
    assert torch.equal(a1, b1) and not torch.equal(a1, a0), "the predictor replayed weight images of the old weights"
    # a training step of the engine on the same model: parameters re-homed into the flat buffer, then updated through raw pointers
    model.train()
    from bench import LOSS_CFG as loss_cfg
    crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), loss_cfg, None, num_modal=2)
    eng = TrainEngine(model, crit, tuple(x.shape))
    lab = (torch.rand((2, 1, 32, 32, 32), generator=torch.Generator().manual_seed(8)) > 0.9).long().to(d)
    eng.step(x, lab)
    model.eval()
    a2, b2 = both()
    assert torch.equal(a2, b2)
    model.train()
    eng.step(x, lab)                                           # same storage, no version bump: only the weights epoch tells
    model.eval()
    a3, b3 = both()
    assert torch.equal(a3, b3) and not torch.equal(a3, a2), "the predictor replayed weight images of the weights before the optimisation step"


def test_sliding_window_with_the_hip_model(golden_dir):
    """end to end: HIP VeloxSeg (eval) as the predictor of both drivers -> identical blended logits; and against the CPU oracle model
    within the logits tolerance of the model parity tests, argmax equal wherever the top-2 margin exceeds that tolerance."""
    import sys
    sys.path.insert(0, golden_dir)
    from recipe import CASES, fill_state_dict
    from oracle import veloxseg_oracle as O
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    from veloxseg_amd.utils import inference_runtime as IR
    cfg, _ = CASES["g2_32_m2"]
    d = dev()
    model = VeloxSeg(**cfg)
    sd = model.state_dict()
    fill_state_dict(sd)
    model.load_state_dict(sd)
    model = model.to(d).eval()
    x = torch.randn((1, 2, 48, 40, 36), generator=torch.Generator().manual_seed(5))
    with torch.inference_mode():
        got, labels = IR.infer_volume(model, x.to(d), (32, 32, 32), 2, 0.5)
        want = SO.sliding_window_inference(x.to(d), (32, 32, 32), 2, IR.Net(model), 0.5)
        got_eager, _ = IR.infer_volume(model, x.to(d), (32, 32, 32), 2, 0.5, taped=False)
    assert torch.equal(got, want)                       # infer_volume replays the captured forward of a window batch (engine.TapedPredictor) ...
    assert torch.equal(got, got_eager)                  # ... which is bit for bit the eager forward
    # two window batches in flight (the default) == one after the other: the blending keeps the window order
    assert IR.SW_PIPELINE
    IR.SW_PIPELINE = False
    try:
        with torch.inference_mode():
            got_seq, labels_seq = IR.infer_volume(model, x.to(d), (32, 32, 32), 2, 0.5)
    finally:
        IR.SW_PIPELINE = True
    assert torch.equal(got, got_seq) and torch.equal(labels, labels_seq)
    assert any(k != "sig" and e is not None for k, e in model.__dict__["_vx_taped_predictor"].__dict__.get("_replicas", {}).items()), "no replica of the tape was replayed"
    tp = model.__dict__["_vx_taped_predictor"]          # the predictor hangs on the model (no process-wide cache)
    assert any(e is not None and e[2].n_kernels > 0 for e in tp._tapes.values()), "the window batches were expected to replay a launch tape"
    # re-homed parameters (what engine.FlatParams does, or model.to()): the tape holds the OLD addresses and must be re-captured, not replayed
    with torch.no_grad():
        for p_ in model.parameters():
            p_.data = (p_.data * 1.5).clone()
    with torch.inference_mode():
        got2, _ = IR.infer_volume(model, x.to(d), (32, 32, 32), 2, 0.5)
        want2, _ = IR.infer_volume(model, x.to(d), (32, 32, 32), 2, 0.5, taped=False)
    assert torch.equal(got2, want2) and not torch.equal(got2, got), "the taped predictor kept reading the parameters' old storage"
    with torch.no_grad():
        for p_ in model.parameters():
            p_.data = (p_.data / 1.5).clone()
    ocfg = O.OracleConfig(**cfg)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    with torch.no_grad():
        ref = SO.sliding_window_inference(x, (32, 32, 32), 2, lambda w: O.forward(w, sd_cpu, ocfg, training=False), 0.5)
    err = (got.cpu() - ref).abs()
    tol = 1e-4 * ref.abs().clamp(min=1.0)
    assert bool((err <= tol).all()), float(err.max())
    top2 = ref.topk(2, dim=1).values
    sure = (top2[:, 0] - top2[:, 1]) > 2e-4
    assert bool((labels.cpu()[:, 0][sure] == ref.argmax(1)[sure]).all())
    assert float(sure.float().mean()) > 0.99


def test_sliding_window_baseline_volume_240x240x155():
    """BASELINE configs[4]: the full 4 x 240 x 240 x 155 BraTS volume, roi 128^3, overlap 0.5, sw_batch 2 (18 windows, MONAI's clamped last window)
    with the brats128 HIP model as the predictor of BOTH drivers: blended logits bit-identical, fused arg-max labels equal, Dice of the labels
    against themselves through the on-device confusion matrix = 1; window count as SURVEY states."""
    from bench import WORKLOADS
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    from veloxseg_amd.utils import inference_runtime as IR
    cfg, _ = WORKLOADS["brats128"]
    d = dev()
    torch.manual_seed(7)
    model = VeloxSeg(**cfg).to(d).eval()
    x = torch.randn((1, 4, 240, 240, 155), generator=torch.Generator().manual_seed(11)).to(d)
    roi = (128, 128, 128)
    assert len(SO.dense_patch_starts((240, 240, 155), roi, SO.get_scan_interval((240, 240, 155), roi, 0.5))) == 18       # 3 x 3 x 2 windows
    with torch.inference_mode():
        got, labels = IR.infer_volume(model, x, roi, 2, 0.5)
        want = SO.sliding_window_inference(x, roi, 2, IR.Net(model), 0.5)
    assert got.shape == (1, cfg["n_classes"], 240, 240, 155) and bool(torch.isfinite(got).all())
    assert torch.equal(got, want)
    assert torch.equal(labels.long(), want.argmax(1, keepdim=True))


def test_metrics_match_reference_goldens(golden_dir):
    from veloxseg_amd.utils.metric import metrics as M, metrics_brats as MB
    d = dev()
    G = torch.load(os.path.join(golden_dir, "metrics.pt"))
    for case in G["binary"]:
        for dt in (torch.uint8, torch.int64):
            got = M.metrics_tensor(case["gt"].to(d).to(dt), case["pred"].to(d).to(dt))
            for a, b in zip(got, case["metrics_tensor"]):
                assert abs(a - b) <= 1e-7 * max(1.0, abs(b)), (got, case["metrics_tensor"])
        res, string = M.show_deep_metrics([case["logits"].to(d)], case["gt"].to(d).long(), True)
        assert string == case["show"][1], (string, case["show"][1])
        assert all(abs(a - b) <= 1e-7 for a, b in zip(res, case["show"][0]))
    for case in G["brats"]:
        got = MB.cal_dice(case["pred"].to(d), case["gt"].to(d).long())
        for a, b in zip(got, case["cal_dice"]):
            assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (got, case["cal_dice"])
        d1 = float(MB.Dice((case["pred"] == 1).float().to(d), (case["gt"] == 1).float().to(d)))
        assert abs(d1 - case["dice_class1"]) <= 1e-6
        res, string = MB.show_deep_metrics([case["logits"].to(d), -case["logits"].to(d)], case["gt"].to(d).long(), True)
        assert string == case["show"][1], (string, case["show"][1])

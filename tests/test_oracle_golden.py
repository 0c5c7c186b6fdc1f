"""Pin the CPU oracle (oracle/veloxseg_oracle.py) against outputs of the REFERENCE itself.

Fixtures were produced by tests/golden/make_golden.py importing /root/reference (with a MONAI
stand-in) in the build container; this file never touches /root/reference.
"""
import os

import pytest
import torch

from oracle import veloxseg_oracle as O
from recipe import CASES, LOSS_CFG, check_compact, fill_state_dict, make_inputs, sd_sha, tensor_sha, unpack_mask

ATOL, RTOL = 2e-4, 2e-4


def load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


def _case(golden_dir, name):
    fix = load(golden_dir, name + ".pt")
    cfg_d, B = CASES[name]
    cfg = O.OracleConfig(**cfg_d)
    sd = fill_state_dict(O.state_dict_template(cfg), seed=fix["sd_seed"])
    assert list(sd.keys()).sort() == list(fix["sd_keys"]).sort()
    assert set(sd.keys()) == set(fix["sd_keys"]), set(sd.keys()) ^ set(fix["sd_keys"])
    for k in sd:
        assert list(sd[k].shape) == fix["sd_shapes"][k], k
    assert sd_sha(sd) == fix["sd_sha256"], "state-dict recipe drifted (RNG?)"
    x, labels = make_inputs(cfg_d, B)
    assert tensor_sha(x) == fix["x_sha256"] and tensor_sha(labels) == fix["labels_sha256"]
    return fix, cfg, sd, x, labels


@pytest.mark.parametrize("name", list(CASES))
def test_eval_logits_and_argmax(golden_dir, name):
    fix, cfg, sd, x, labels = _case(golden_dir, name)
    with torch.no_grad():
        logits = O.forward(x, sd, cfg, training=False)
    check_compact(logits, fix["eval_logits"], ATOL, RTOL, "eval logits")
    am = logits.argmax(1).to(torch.uint8)
    mism = (am != unpack_mask(fix["argmax"])).float().mean().item()
    assert mism == 0.0, f"argmax differs on {mism:.2e} of voxels"


@pytest.mark.parametrize("name", ["g1_48_m2", "g3_64_brats", "g4_aniso_m2", "g5_128_m2", "g7_96_m2"])
def test_train_outputs_loss_grads(golden_dir, name):
    fix, cfg, sd, x, labels = _case(golden_dir, name)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v)}
    full = dict(sd)
    full.update(params)
    outs = O.forward(x, full, cfg, training=True)
    assert len(outs) == len(fix["train_outputs"]) == 4 + 1 + 1 + cfg.M
    for i, (o, r) in enumerate(zip(outs, fix["train_outputs"])):
        check_compact(o, r, ATOL, RTOL, f"train output {i}")
    L = O.loss(outs, labels, x, cfg.M, LOSS_CFG)
    assert abs(float(L) - fix["loss"]) <= 1e-4 * abs(fix["loss"])
    L.backward()
    bad = []
    for k, p in params.items():
        gn = float(p.grad.double().norm())
        ref = fix["grad_norms"][k]
        if abs(gn - ref) > 2e-3 * max(ref, 1e-3):
            bad.append((k, gn, ref))
    assert not bad, bad[:5]
    for k, g in fix["grads_small"].items():
        torch.testing.assert_close(params[k].grad, g, atol=5e-4, rtol=5e-3, msg=lambda m: f"{k}: {m}")


def test_runtime_helper_known_answers(golden_dir):
    """reference tests/test_runtime_helpers.py:63-75,87-111 restated against the oracle helpers."""
    assert O.normalized_deep_loss_weights([1, 1, 1, 1], 5) == [0.2] * 5
    with pytest.raises(ValueError, match="deep_Loss_weight"):
        O.normalized_deep_loss_weights([4, 2, 1], 5)
    with pytest.raises(ValueError, match="sum"):
        O.normalized_deep_loss_weights([0, 0, 0, 0], 5)
    assert O.veloxseg_output_layout(8, 2) == {"seg": (0, 4), "reconstruction": 4, "decoder_gram": 5, "teacher_grams": (6, 7)}
    assert O.veloxseg_output_layout(5, 2) == {"seg": (0, 1), "reconstruction": 1, "decoder_gram": 2, "teacher_grams": (3, 4)}
    with pytest.raises(ValueError, match="VeloxSeg"):
        O.veloxseg_output_layout(4, 2)
    ops = load(golden_dir, "ops.pt")["runtime"]
    assert O.normalized_deep_loss_weights([4, 2, 1, 1], 4) == ops["w4"]
    assert O.veloxseg_output_layout(8, 2) == ops["layout8"]


def test_micro_ops(golden_dir):
    ops = load(golden_dir, "ops.pt")
    g = ops["layernorm"]
    torch.testing.assert_close(O.layernorm_cf(g["x"], g["w"], g["b"]), g["y"], atol=1e-5, rtol=1e-5)
    # gather / scatter / full attention module (M=2)
    g = ops["pwa"]
    plan = O.plan_pwa([12, 12, 12], [6, 6, 6], [1, 1, 1], 2, 2, 8, 32)
    assert plan["ch_qk"] == g["channels_qk"] and plan["ch_v"] == g["channels_v"] and plan["big"] == g["big"]
    tok = O.gather_windows(g["q"], plan, plan["c_qk"])
    torch.testing.assert_close(tok, g["tok"], atol=0, rtol=0)
    torch.testing.assert_close(O.scatter_windows(g["tok"], plan, plan["c_qk"]), g["scat"], atol=1e-5, rtol=1e-5)
    xs = [t.clone().requires_grad_(True) for t in g["xs"]]
    sd = {k: (v.clone().requires_grad_(True) if torch.is_floating_point(v) else v) for k, v in g["sd"].items()}
    ys = O.pwa_attention_module(xs, sd, "", plan, dict(attn=0.0, proj=0.0), False)
    for y, r in zip(ys, g["ys"]):
        torch.testing.assert_close(y, r, atol=2e-5, rtol=2e-5)
    torch.autograd.backward(ys, g["gy"])
    for x, r in zip(xs, g["gxs"]):
        torch.testing.assert_close(x.grad, r, atol=5e-5, rtol=1e-4)
    for k, r in g["gparams"].items():
        torch.testing.assert_close(sd[k].grad, r, atol=2e-4, rtol=1e-3, msg=lambda m: f"{k}: {m}")
    # block: double residual
    g = ops["block"]
    plan = O.plan_pwa([8, 8, 8], [2, 2, 2], [1, 1, 1], 2, 1, 4, 16)
    assert plan["nb"] == 3 and plan["ch_v"] == g["channels_v"]
    y = O.pwa_block([g["x"]], g["sd"], "", plan, dict(attn=0.0, proj=0.0), False)[0]
    torch.testing.assert_close(y, g["y"], atol=2e-5, rtol=2e-5)
    g = ops["patchmerge"]
    torch.testing.assert_close(O.patch_merging(g["x"], g["sd"], ""), g["y"], atol=2e-5, rtol=2e-5)
    g = ops["jlc"]
    torch.testing.assert_close(O.jlc(g["x"], g["sd"], "", 4, 0.0, False), g["y"], atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(O.down_conv(ops["down4"]["x"], ops["down4"]["sd"], "", 4), ops["down4"]["y"], atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(O.down_conv(ops["down2"]["x"], ops["down2"]["sd"], "", 2), ops["down2"]["y"], atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(O.up_conv(ops["up2"]["x"], ops["up2"]["sd"], ""), ops["up2"]["y"], atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(O.pixel_shuffle3d(ops["pixelshuffle"]["x"], 4), ops["pixelshuffle"]["y"], atol=0, rtol=0)
    torch.testing.assert_close(O.gram(ops["gram"]["x"]), ops["gram"]["y"], atol=1e-6, rtol=1e-5)
    for ncls in (2, 4):
        g = ops[f"segloss{ncls}"]
        lab = g["lab"].long()
        assert abs(float(O.dice_loss(g["logit"], lab)) - g["dice"]) < 1e-6
        lg = g["logit"].clone().requires_grad_(True)
        O.seg_loss(lg, lab).backward()
        torch.testing.assert_close(lg.grad, g["glogit"], atol=1e-8, rtol=1e-4)
    g = ops["loss_full"]
    outs = [o.clone().requires_grad_(True) for o in g["outs"]]
    L = O.loss(outs, g["lab"].long(), g["sr"], 2, g["cfg"])
    assert abs(float(L) - g["loss"]) < 1e-5
    L.backward()
    for o, r in zip(outs, g["gouts"]):
        torch.testing.assert_close(o.grad, r, atol=1e-8, rtol=1e-4)


def test_upsample_matches_aten():
    x = torch.randn(1, 3, 4, 6, 3)
    y = O.upsample_trilinear(x, [16, 24, 12])
    r = torch.nn.functional.interpolate(x, size=[16, 24, 12], mode="trilinear", align_corners=True)
    torch.testing.assert_close(y, r, atol=2e-6, rtol=1e-5)

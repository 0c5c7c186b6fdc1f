"""bf16 opt-in mode (BASELINE configs[1]: BraTS2021 128^3 bf16; functional.set_precision("bf16")): bf16 MFMA operands with fp32 accumulation in the
patch-expand layers, fp32 storage and fp32 everything else.  The mode is NOT held to the fp32 logit tolerance; its gate is SURVEY's: Dice delta
< 1e-3 against the fp32 path of this library (whose parity with the reference the other tests establish), here per class on the BASELINE-size
workloads, plus op-level error bounds of bf16 rounding (2^-9 per operand, fp32 sums) and a training run that tracks the fp32 one."""
import os
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(autouse=True)
def _fp32_afterwards():
    yield
    from veloxseg_amd import functional as VF
    VF.set_precision("fp32")


@pytest.mark.parametrize("Cout", [64, 128])
def test_patch_expand_bf16_operands_stay_within_bf16_rounding(Cout):
    from veloxseg_amd import functional as VF
    d = torch.device("cuda:0")
    g = torch.Generator(device=d).manual_seed(5)
    x = torch.randn(2, 16, 8, 8, 16, device=d, generator=g)
    w = torch.randn(Cout, 16, 3, 3, 3, device=d, generator=g) * (16 * 27) ** -0.5
    b = torch.randn(Cout, device=d, generator=g) * 0.1
    res = {}
    for mode in ("fp32", "bf16"):
        VF.set_precision(mode)
        assert VF.get_precision() == mode
        xx, ww = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = VF.conv3d(xx, ww, b, stride=1, padding=1, pixel_shuffle=4)
        gy = torch.randn(y.shape, device=d, generator=torch.Generator(device=d).manual_seed(1))
        y.backward(gy)
        res[mode] = (y.detach(), xx.grad.clone(), ww.grad.clone())
    for i, what in enumerate(("output", "input gradient", "weight gradient")):      # (round 3: the weight gradient runs with bf16 operands too)
        a, r = res["bf16"][i], res["fp32"][i]
        rel = float((a - r).norm() / r.norm())
        assert 1e-4 < rel < 5e-3, (what, rel)            # really bf16 operands (not the fp32 kernel), and no worse than their rounding
        assert float((a - r).abs().max()) < 1e-2 * float(r.abs().max()), what


@pytest.mark.parametrize("workload", ["brats128", "autopet128"])
def test_bf16_mode_dice_delta_at_baseline_size(workload):
    """BASELINE configs[1] (BraTS 128^3, M = 1 with 4 channels, 4 classes) and the headline workload: arg-max masks of the bf16 mode against the fp32
    mode on the same weights and volume: per-class Dice against the labels moves by < 1e-3 (SURVEY 8c gate for bf16) and the two masks agree to
    Dice > 0.995."""
    from bench import WORKLOADS, synth
    from veloxseg_amd import functional as VF
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    cfg, _ = WORKLOADS[workload]
    x, lab = synth(cfg, 1, "cuda", 12345)
    outs = {}
    for mode in ("fp32", "bf16"):
        VF.set_precision(mode)
        torch.manual_seed(3)
        model = VeloxSeg(**cfg).cuda().eval()
        with torch.no_grad():
            outs[mode] = model(x)
    a, r = outs["bf16"], outs["fp32"]
    assert float((a - r).abs().max()) > 0, "the bf16 mode must actually take the bf16 kernels at this size"
    am, rm, gt = a.argmax(1), r.argmax(1), lab[:, 0]
    assert float((am != rm).float().mean()) < 5e-3

    def dice(p, q, c):
        return 2.0 * float(((p == c) & (q == c)).sum()) / max(1.0, float((p == c).sum() + (q == c).sum()))
    for c in range(1, a.shape[1]):
        assert abs(dice(am, gt, c) - dice(rm, gt, c)) < 1e-3, (c, dice(am, gt, c), dice(rm, gt, c))
        assert dice(am, rm, c) > 0.995, (c, dice(am, rm, c))


def test_bf16_training_tracks_fp32_training():
    """TrainEngine(precision="bf16") (launch tapes captured with the bf16 kernels) against the fp32 engine: same data, same seeds, 4 AdamW steps"""
    from bench import LOSS_CFG, WORKLOADS, synth
    from veloxseg_amd import functional as VF
    from veloxseg_amd.engine import TrainEngine
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    from veloxseg_amd.utils.loss import Loss
    cfg, _ = WORKLOADS["brats128"]
    crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg["in_ch"]))
    x, lab = synth(cfg, 1, "cuda", 12345)
    losses = {}
    for mode in ("fp32", "bf16"):
        VF.reset_dropout_sites()
        torch.manual_seed(12345)
        model = VeloxSeg(**cfg).cuda()
        VF.manual_seed(5, "cuda")
        eng = TrainEngine(model, crit, (1, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=True, overlap=False, precision=mode)
        losses[mode] = [float(eng.step(x, lab)) for _ in range(4)]
        assert eng.use_graph, "tape self-check failed"
        # the engine's precision / fork / RNG settings are its own: nothing process-wide is left behind by its capture or its steps
        assert VF.get_precision() == "fp32" and VF.RNG_INPLACE is False, (VF.get_precision(), VF.RNG_INPLACE)
        del eng, model
    for a, r in zip(losses["bf16"], losses["fp32"]):
        assert abs(a - r) <= 1e-2 * abs(r), losses
    assert losses["bf16"][-1] < losses["bf16"][0]


# ---- round 6: the bf16 STORAGE mode (16-bit tensors in HBM, not only 16-bit matrix-pipe operands) --------------------------------------------------------------
def _rel(a, r):
    a, r = a.detach().double().cpu(), r.detach().double().cpu()
    return float((a - r).norm() / r.norm().clamp_min(1e-30))


@pytest.mark.parametrize("case", [("L1_32cube", 2, 16, 4, 3, (32, 32, 32)), ("L1_aniso", 1, 16, 4, 3, (8, 12, 20)), ("hecktor_L1", 1, 16, 4, 3, (32, 32, 16))], ids=lambda c: c[0])
def test_jlc_block_with_16bit_internal_tensors_vs_oracle(case):
    """JLC block (conv_blocks.py:41-75) of the 32^3 level in the bf16 storage mode: y_k, o, dn, d_o, g_k are bf16 arrays, bf16 MFMA operands.  Against the fp32 CPU
    oracle: relative RMS error of the output and of every gradient <= 1e-2 (bf16 rounding = 2^-9 per stored element / operand, fp32 sums); and the 16-bit storage must
    not be much worse than the operand-only mode of rounds 2-5 (VELOXSEG_BF16_STORAGE=0), which shares its matrix-pipe arithmetic."""
    from oracle import veloxseg_oracle as O
    from test_fused_blocks_gpu import _jlc_module, _oracle_sd
    from veloxseg_amd import functional as VF
    _, B, C, G, r, sp = case
    cm = VF.cpp_module()
    x = torch.randn(B, C, *sp, generator=torch.Generator().manual_seed(5))
    gy = None
    res = {}
    for storage in (True, False):
        VF.BF16_STORAGE = storage
        try:
            VF.set_precision("bf16")
            assert cm.get_act_bf16() == storage
            m = _jlc_module(C, G, r, 0.0, 3).cuda().train()
            xg = x.cuda().requires_grad_(True)
            out = m(xg)
            if gy is None:
                gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(6))
            out.backward(gy.cuda())
            torch.cuda.synchronize()
            res[storage] = (out.detach().cpu(), xg.grad.cpu(), {k: p.grad.cpu().clone() for k, p in m.named_parameters()}, m)
        finally:
            VF.BF16_STORAGE = True
            VF.set_precision("fp32")
    m = res[True][3]
    sd = _oracle_sd(m)
    xc = x.clone().requires_grad_(True)
    ref = O.jlc(xc, sd, "blk.", G, 0.0, True)
    ref.backward(gy)
    errs = {}
    for storage in (True, False):
        out, dx, gr, _ = res[storage]
        e = {"out": _rel(out, ref), "dx": _rel(dx, xc.grad)}
        for k, g in gr.items():
            if "spatial_convs" in k and k.endswith("bias"):
                continue          # behind an InstanceNorm: zero by construction, not computed
            e["d" + k] = _rel(g, sd["blk." + k].grad)
        errs[storage] = e
    assert float((res[True][0] - res[False][0]).abs().max()) > 0, "the storage mode must take the 16-bit kernels"
    for k, v in errs[True].items():
        assert v <= 1e-2, (k, v, errs)
        assert v <= 3.0 * errs[False][k] + 2e-3, (k, v, errs[False][k])


@pytest.mark.parametrize("name,B", [("g6_128_brats", 2), ("g5_128_m2", 2), ("g7_96_m2", 2)])
def test_bf16_storage_mode_taped_step_vs_oracle(name, B):
    """BASELINE configs[1] (BraTS 128^3, batch 2) and the headline model (M = 2, 128^3; and the shipped 96^3 geometry) in the bf16 STORAGE mode -- 16-bit full-resolution
    heads / reconstructions and their gradients, 16-bit block-internal tensors of the 32^3-level JLC blocks, bf16 matrix-pipe operands; fp32 statistics, sums, soft-max,
    loss, master weights and flat gradients (reference speed_test.py:122,127: torch.amp.autocast) -- through TrainEngine's launch tapes (the path bench.py --dtype bf16
    times), against the fp32 CPU ORACLE on the same weights and batch (dropout 0, as every oracle comparison).  Gates: logits relative RMS <= 1e-2,
    per-class Dice of the arg-max masks against the labels within 1e-3 of the oracle's, loss within 1 %; and the flat gradient within 3e-2 relative RMS."""
    from oracle import veloxseg_oracle as O
    from recipe import CASES, LOSS_CFG, fill_state_dict, make_inputs
    from veloxseg_amd import functional as VF
    from veloxseg_amd.engine import TrainEngine
    from veloxseg_amd.model.VeloxSeg import VeloxSeg
    from veloxseg_amd.utils.loss import Loss
    cfg_d, _ = CASES[name]
    model = VeloxSeg(**cfg_d)
    sd = fill_state_dict(model.state_dict(), seed=7)
    model.load_state_dict(sd)
    model = model.cuda().train()
    x, labels = make_inputs(cfg_d, B)
    cfg = O.OracleConfig(**cfg_d)
    crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, torch.device("cuda"), num_modal=cfg.M)
    eng = TrainEngine(model, crit, (B, sum(cfg_d["in_ch"]), *cfg_d["input_size"]), lr=0.0, weight_decay=0.0, use_graph=True, overlap=False, precision="bf16")
    eng.step(x.cuda(), labels.cuda())
    torch.cuda.synchronize()
    assert eng.use_graph and eng.graphs is not None, "the capture's self-check failed: the step did not run as launch tapes"
    loss = float(eng.step())                 # a REPLAYED step (lr = 0: same parameters)
    torch.cuda.synchronize()
    outs = eng.last_outputs
    assert outs[0].dtype == torch.bfloat16, "the full-resolution logits must be a 16-bit tensor in this mode"
    assert outs[4].dtype == torch.bfloat16, "the reconstructions must be 16-bit tensors in this mode"
    flat = eng.flat.grad.detach().cpu().double()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v)}
    full = dict(sd)
    full.update(params)
    ro = O.forward(x, full, cfg, training=True)
    rl = O.loss(ro, labels, x, cfg.M, LOSS_CFG)
    rl.backward()
    assert abs(loss - float(rl)) <= 1e-2 * abs(float(rl)), (loss, float(rl))
    lg, ref = outs[0].float().cpu(), ro[0].detach()
    rel = float((lg - ref).norm() / ref.norm())
    assert rel <= 1e-2, rel
    rc, rref = outs[4].float().cpu(), ro[4].detach()
    assert float((rc - rref).norm() / rref.norm()) <= 1e-2
    am, amr, gt = lg.argmax(1), ref.argmax(1), labels[:, 0]

    def dice(p, q, c):
        return 2.0 * float(((p == c) & (q == c)).sum()) / max(1.0, float((p == c).sum() + (q == c).sum()))
    for c in range(1, lg.shape[1]):
        assert abs(dice(am, gt, c) - dice(amr, gt, c)) < 1e-3, (c, dice(am, gt, c), dice(amr, gt, c))
    gref = torch.zeros_like(flat)
    for n, p in zip(eng.flat.names, eng.flat.params):
        o, k = eng.flat.slices[n]
        gref[o:o + k] = params[n].grad.double().reshape(-1)
    grel = float((flat - gref).norm() / gref.norm())
    assert grel <= 3e-2, grel
    print(f"[bf16 storage vs oracle] {name} B={B}: loss {loss:.6f} vs {float(rl):.6f}, logits rel RMS {rel:.2e}, arg-max differs on {float((am != amr).float().mean()):.2e}, flat gradient rel RMS {grel:.2e}")

import sys, os, types, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
from recipe import CASES, LOSS_CFG, make_inputs, fill_state_dict
from veloxseg_amd import functional as VF
from veloxseg_amd.model.VeloxSeg import VeloxSeg
import torch.nn.functional as F
d = torch.device("cuda:0")
# op level
torch.manual_seed(0)
for Cout in (64, 128):
    x = torch.randn(2, 16, 8, 8, 16, device=d); w = torch.randn(Cout, 16, 3, 3, 3, device=d) * (16 * 27) ** -0.5; b = torch.randn(Cout, device=d) * 0.1
    res = {}
    for mode in ("fp32", "bf16"):
        VF.set_precision(mode)
        xx = x.clone().requires_grad_(True); ww = w.clone().requires_grad_(True)
        y = VF.conv3d(xx, ww, b, stride=1, padding=1, pixel_shuffle=4)
        g = torch.randn(y.shape, device=d, generator=torch.Generator(device=d).manual_seed(1))
        y.backward(g)
        res[mode] = (y.detach(), xx.grad.clone(), ww.grad.clone())
    for i, n in enumerate(("y", "dx", "dw")):
        a, b_ = res["bf16"][i], res["fp32"][i]
        print(Cout, n, "max abs err", float((a - b_).abs().max()), "ref max", float(b_.abs().max()), "rel rms", float((a - b_).norm() / b_.norm()))
VF.set_precision("fp32")
# model level
for name in ("autopet128", "brats128"):
    from bench import WORKLOADS, synth
    cfg_d, B = WORKLOADS[name]
    x, lab = synth(cfg_d, 1, "cuda", 12345)
    outs = {}
    for mode in ("fp32", "bf16"):
        VF.set_precision(mode)
        torch.manual_seed(3)
        model = VeloxSeg(**cfg_d).cuda().eval()
        with torch.no_grad():
            outs[mode] = model(x)
    a, b_ = outs["bf16"], outs["fp32"]
    am, bm = a.argmax(1), b_.argmax(1)
    ncls = a.shape[1]
    print(name, "logit max abs diff", float((a - b_).abs().max()), "ref max", float(b_.abs().max()), "argmax mismatch frac", float((am != bm).float().mean()))
    for c in range(1, ncls):
        da = 2 * ((am == c) & (lab[:, 0] == c)).sum().item() / max(1, ((am == c).sum() + (lab[:, 0] == c).sum()).item())
        db = 2 * ((bm == c) & (lab[:, 0] == c)).sum().item() / max(1, ((bm == c).sum() + (lab[:, 0] == c).sum()).item())
        agree = 2 * ((am == c) & (bm == c)).sum().item() / max(1, ((am == c).sum() + (bm == c).sum()).item())
        print("  class", c, "dice vs labels bf16", da, "fp32", db, "delta", da - db, "| dice(bf16 mask, fp32 mask)", agree)
VF.set_precision("fp32")

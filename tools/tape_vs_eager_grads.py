#!/usr/bin/env python3
"""Per-parameter comparison of the taped step's gradients against the eager step's, from the same weights and dropout streams.
  python tools/tape_vs_eager_grads.py [workload] [batch]"""
import os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import LOSS_CFG, WORKLOADS, synth
from veloxseg_amd import functional as VF
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss

wl = sys.argv[1] if len(sys.argv) > 1 else "brats128"
cfg, B = WORKLOADS[wl]
if len(sys.argv) > 2:
    B = int(sys.argv[2])
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg["in_ch"]))
x, lab = synth(cfg, B, "cuda", 12345)
eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=True, overlap=False, verify_replays=0)
eng.step(x, lab)            # captures
torch.cuda.synchronize()
assert eng.graphs is not None, "capture failed"
rng = VF.rng_state(eng.dev)
rng0 = rng.clone()
names = [n for n, _ in model.named_parameters()]
res = {}
for mode in ("eager", "tape", "eager2", "tape2", "eager3", "eager4"):
    rng.copy_(rng0)
    torch.cuda.synchronize()
    if mode.startswith("eager"):
        with eng._settings(capture=True):          # the engine's own switches (in-place RNG step, forks) for its eager pass
            eng._eager_pass()
    else:
        eng._replay(comm=False)
    torch.cuda.synchronize()
    res[mode] = (float(eng.loss), eng.flat.grad.clone())
    print(mode, "loss", res[mode][0], "|g|", float(res[mode][1].double().abs().sum()))
ge, gt = res["eager"][1], res["tape"][1]
print("eager vs eager2 max", float((ge - res["eager2"][1]).abs().max()), " tape vs tape2 max", float((gt - res["tape2"][1]).abs().max()))
rows = []
for name, p in model.named_parameters():
    sl = eng.flat.slices.get(name)
    if sl is None:
        continue
    lo, hi = sl[0], sl[0] + sl[1]
    a, b = ge[lo:hi], gt[lo:hi]
    d = float((a - b).abs().max())
    rows.append((d / max(float(a.abs().max()), 1e-12), d, float(a.abs().max()), name, tuple(p.shape)))
rows.sort(reverse=True)
for r in rows[:25]:
    print("rel %.3e  abs %.3e  max|g| %.3e  %s %s" % r)

for other in ("eager2", "eager3", "eager4"):
    go = res[other][1]
    rows = []
    for name, p in model.named_parameters():
        lo, n = eng.flat.slices[name]
        a, b = gt[lo:lo + n], go[lo:lo + n]
        d = float((a - b).abs().max())
        if float(a.abs().max()) > 1e-6:
            rows.append((d / float(a.abs().max()), d, float(a.abs().max()), name, tuple(p.shape)))
    rows.sort(reverse=True)
    print("--", other, "vs tape")
    for r in rows[:12]:
        print("rel %.3e  abs %.3e  max|g| %.3e  %s %s" % r)

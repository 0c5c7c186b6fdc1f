#!/usr/bin/env python3
"""Repeat the eager training pass from the same weights / dropout streams and print |grad| of every pass (a race shows up as a pass that differs).
  python tools/eager_flake.py [workload] [batch] [graph:0|1] [passes]"""
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
graph = len(sys.argv) > 3 and sys.argv[3] == "1"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 12
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg["in_ch"]))
x, lab = synth(cfg, B, "cuda", 12345)
eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=graph, overlap=False, verify_replays=0)
eng.step(x, lab)
torch.cuda.synchronize()
rng = VF.rng_state(eng.dev)
rng0 = rng.clone()
ref = None
side = torch.cuda.Stream()
for i in range(n):
    rng.copy_(rng0)
    torch.cuda.synchronize()
    if os.environ.get("VX_ON_SIDE", "0") == "1":
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with eng._settings(capture=True):          # the engine's own switches (in-place RNG step, forks) for its eager pass
                eng._eager_pass()
    else:
        with eng._settings(capture=True):          # the engine's own switches (in-place RNG step, forks) for its eager pass
            eng._eager_pass()
    torch.cuda.synchronize()
    g = eng.flat.grad.clone()
    if ref is None:
        ref = g
    print(i, "loss %.8f |g| %.6f  max diff vs pass 0: %.3e" % (float(eng.loss), float(g.double().abs().sum()), float((g - ref).abs().max())))

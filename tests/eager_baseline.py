#!/usr/bin/env python3
"""Ad-hoc measurement (NOT part of the product or of bench.py): the oracle's plain PyTorch formulation run with
aten/MIOpen kernels on the MI355X = "PyTorch-ROCm eager" training step, the reference point of BASELINE.json's
">= 5x eager" target.  Usage: python tests/eager_baseline.py [workload] [batch] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402
from bench import LOSS_CFG, WORKLOADS, synth  # noqa: E402
from oracle import veloxseg_oracle as O  # noqa: E402
from recipe import fill_state_dict  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "autopet128"
cfg, defB = WORKLOADS[wl]
B = int(sys.argv[2]) if len(sys.argv) > 2 else defB
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda")
ocfg = O.OracleConfig(**{**cfg, "attn_drop": 0.1})
sd = {k: v.to(dev) for k, v in fill_state_dict(O.state_dict_template(ocfg), seed=7).items()}
params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if torch.is_floating_point(v)}
full = dict(sd)
full.update(params)
opt = torch.optim.AdamW(list(params.values()), lr=2.5e-4, weight_decay=0.01)
x, lab = synth(cfg, B, dev, 12345)


def step():
    opt.zero_grad(set_to_none=True)
    outs = O.forward(x, full, ocfg, training=True)
    loss = O.loss(outs, lab, x, ocfg.M, LOSS_CFG)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"eager {wl} B={B}: {dt * 1e3:.2f} ms/step = {B / dt:.2f} patches/s (loss {float(l):.4f})")

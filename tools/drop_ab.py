#!/usr/bin/env python3
"""What do the dropout masks cost the taped step?  The same workload with proj / conv / attention dropout switched off (timing only: the arithmetic differs)."""
import os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import LOSS_CFG, WORKLOADS, synth
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss

wl = sys.argv[1] if len(sys.argv) > 1 else "autopet128"
cfg0, B = WORKLOADS[wl]
cfg0 = dict(cfg0)
cfg0.setdefault("attn_drop", 0.1)          # (the model's default, not spelled out in bench.WORKLOADS)
keys = [k for k in cfg0 if "drop" in k]
print("dropout keys:", {k: cfg0[k] for k in keys})
for label, off in (("as configured", []), ("attention dropout off", [k for k in keys if "attn" in k]), ("proj / conv dropout off", [k for k in keys if "attn" not in k]), ("all off", keys)):
    cfg = dict(cfg0)
    for k in off:
        cfg[k] = 0.0
    torch.manual_seed(12345)
    model = VeloxSeg(**cfg).cuda()
    crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg["in_ch"]))
    x, lab = synth(cfg, B, "cuda", 12345)
    eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=True, overlap=False, pipeline_tail=True)
    for _ in range(20):
        eng.step(x, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        eng.step(x, lab)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    print(f"{label:28s} {ms:6.3f} ms per step  {B / ms * 1e3:7.1f} patches/s")
    del eng, model
    torch.cuda.empty_cache()

"""In-process A/B of a runtime knob on the bench workload (box-to-box and run-to-run clock differences cancel): alternates the settings
every 20 steps for several rounds and prints the mean ms/step of each.  argv[1] = knob: fuse_gelu"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloxseg_amd  # noqa
import torch
from veloxseg_amd import functional as VF
from bench import WORKLOADS, LOSS_CFG, synth
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss

knob = sys.argv[1] if len(sys.argv) > 1 else "fuse_gelu"
cfg, B = WORKLOADS["autopet128"]
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=2)
eng = TrainEngine(model, crit, (B, 2, 128, 128, 128))
x, lab = synth(cfg, B, "cuda", 12345)
for _ in range(5): eng.step(x, lab)
m = VF.cpp_module()


def setting(v):
    if knob == "fuse_gelu":
        m.set_fuse_gelu(bool(v))
    else:
        raise SystemExit("unknown knob")


acc = {0: [], 1: []}
for rnd in range(6):
    for v in (1, 0):
        setting(v)
        for _ in range(3): eng.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): eng.step()
        torch.cuda.synchronize(); acc[v].append((time.perf_counter() - t0) / 20 * 1e3)
setting(1)
for v in (1, 0):
    print(knob, "=", v, " ms/step per round:", " ".join("%.2f" % t for t in acc[v]), " mean %.3f" % (sum(acc[v]) / len(acc[v])))

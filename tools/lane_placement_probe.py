#!/usr/bin/env python3
"""Is the occasional slow process (about one run in ten lands 9-18 % lower with all four lanes reported distinct) a matter of WHICH calibrated stream serves
which tape lane?  In one process: steady-state step time under several lane permutations.  argv: [workload]"""
import ctypes, itertools, os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import LOSS_CFG, WORKLOADS, synth
from veloxseg_amd import _hip as H
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss

wl = sys.argv[1] if len(sys.argv) > 1 else "autopet128"
cfg, B = WORKLOADS[wl]
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg["in_ch"]))
x, lab = synth(cfg, B, "cuda", 12345)
eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]), use_graph=True, overlap=False)
for _ in range(5):
    eng.step(x, lab)
torch.cuda.synchronize()


def ms(n=25):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.step(x, lab)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = [("identity", ms())]
for perm in list(itertools.permutations(range(4)))[1:]:
    arr = (ctypes.c_int * 4)(*perm)
    H.call("vx_tape_permute_lanes", ctypes.addressof(arr))
    eng.__dict__.pop("_lane_cache", None)
    for _ in range(3):
        eng.step(x, lab)
    out.append((str(perm), ms()))
    inv = [0] * 4
    for k, p_ in enumerate(perm):
        inv[p_] = k
    arr = (ctypes.c_int * 4)(*inv)
    H.call("vx_tape_permute_lanes", ctypes.addressof(arr))          # back to the original assignment
    eng.__dict__.pop("_lane_cache", None)
print(wl, "lane_on_caller_queue", H.query("vx_tape_lane_on_caller_queue"), " ".join(f"{k}:{v:.3f}" for k, v in sorted(out, key=lambda kv: kv[1])[:8]), flush=True)

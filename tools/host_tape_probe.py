#!/usr/bin/env python3
"""Host cost of one taped step: (a) enqueue of ONE step on an idle GPU (no back-pressure from full hardware queues), wall and process CPU time; (b) enqueue and completion
of 20 steps back to back.  If (a) is close to the step time of (b), the step is bound by the launch path, not by the GPU.  argv: [workload]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS, LOSS_CFG, synth
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
for _ in range(10):
    eng.step(x, lab)
torch.cuda.synchronize()
one = []
for _ in range(30):
    torch.cuda.synchronize()
    time.sleep(0.002)
    t0 = time.perf_counter(); c0 = time.process_time()
    eng.step(x, lab)
    t1 = time.perf_counter(); c1 = time.process_time()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    one.append(((t1 - t0) * 1e3, (c1 - c0) * 1e3, (t2 - t0) * 1e3))
one.sort()
med = one[len(one) // 2]
print("one step from idle: enqueue %.3f ms wall (%.3f ms process CPU), finished after %.3f ms   [median of 30; min enqueue %.3f]" % (med[0], med[1], med[2], one[0][0]))
for rnd in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); c0 = time.process_time()
    for _ in range(20):
        eng.step(x, lab)
    t1 = time.perf_counter(); c1 = time.process_time(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("20 steps: enqueue %.3f ms/step (process CPU %.3f), finished %.3f ms/step, GPU tail after the last enqueue %.3f ms"
          % ((t1 - t0) / 20 * 1e3, (c1 - c0) / 20 * 1e3, (t2 - t0) / 20 * 1e3, (t2 - t1) * 1e3))
print("load average:", os.getloadavg(), "cpus:", os.cpu_count())

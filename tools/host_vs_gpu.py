"""Is the eager step host-bound or GPU-bound?  For several rounds: time to ENQUEUE 20 steps (no sync) and time until the GPU has finished them."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloxseg_amd  # noqa
import torch
from bench import WORKLOADS, LOSS_CFG, synth
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss
wl = sys.argv[1] if len(sys.argv) > 1 else "autopet128"
cfg, B = WORKLOADS[wl]
B = int(sys.argv[2]) if len(sys.argv) > 2 else B
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=len(cfg["in_ch"]))
eng = TrainEngine(model, crit, (B, sum(cfg["in_ch"]), *cfg["input_size"]))
x, lab = synth(cfg, B, "cuda", 12345)
for _ in range(5): eng.step(x, lab)
for rnd in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter(); c0 = time.process_time()
    for _ in range(20): eng.step()
    t1 = time.perf_counter(); c1 = time.process_time(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("round %d: enqueue %.2f ms/step (process CPU time %.2f), finished %.2f ms/step, GPU tail after the last enqueue %.2f ms"
          % (rnd, (t1 - t0) / 20 * 1e3, (c1 - c0) / 20 * 1e3, (t2 - t0) / 20 * 1e3, (t2 - t1) * 1e3))
print("load average:", os.getloadavg(), "cpus:", os.cpu_count())

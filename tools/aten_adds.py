"""Where do the aten `add` / `fill` launches of one training step come from?  Runs the engine's staged pass eagerly under torch.profiler (CPU activity, shapes) and prints every
aten::add / add_ / fill_ / zero_ / zeros_like with its input shapes and the autograd node (evaluate_function) it ran under."""
import os, sys, types, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import WORKLOADS, LOSS_CFG, synth
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss
cfg, B = WORKLOADS["autopet128"]
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=2)
eng = TrainEngine(model, crit, (B, 2, 128, 128, 128), use_graph=True)
x, lab = synth(cfg, B, "cuda", 12345)
for _ in range(3):
    eng.step(x, lab)
torch.cuda.synchronize()
with eng._settings(capture=True):
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
        eng._eager_pass()
        torch.cuda.synchronize()
evs = sorted(prof.events(), key=lambda e: e.time_range.start)
parents = [e for e in evs if e.name.startswith("autograd::engine::evaluate_function") or e.name.endswith("Backward") or "Backward" in e.name]
cnt = collections.Counter()
for e in evs:
    if e.name in ("aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::zeros_like", "aten::zeros", "aten::sum", "aten::mul", "aten::copy_", "aten::clone", "aten::contiguous"):
        par = [p.name for p in parents if p.time_range.start <= e.time_range.start and p.time_range.end >= e.time_range.end]
        cnt[(e.name, str(e.input_shapes)[:80], (par[0] if par else "-")[:90])] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, k)

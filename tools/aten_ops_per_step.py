"""Which aten operators (torch-native launches, not C-ABI entries) run in one training step, with their python call sites: argv = op substring filter"""
import os, sys, types, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloxseg_amd  # noqa
import torch
from torch.profiler import profile, ProfilerActivity
from bench import WORKLOADS, LOSS_CFG, synth
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cfg, B = WORKLOADS["autopet128"]
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=2)
eng = TrainEngine(model, crit, (B, 2, 128, 128, 128))
x, lab = synth(cfg, B, "cuda", 12345)
for _ in range(3): eng.step(x, lab)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    eng.step()
    torch.cuda.synchronize()
cnt = collections.Counter()
sites = collections.defaultdict(collections.Counter)
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    cnt[e.name] += 1
    if flt and flt in e.name:
        st = [s for s in (e.stack or []) if "veloxseg_amd" in s or "autograd" in s][:3]
        sites[e.name][" <- ".join(s.split("/")[-1] for s in st)] += 1
for k, v in cnt.most_common(40):
    print("%5d  %s" % (v, k))
for k, c in sites.items():
    print("==", k)
    for s, v in c.most_common(25):
        print("   %4d  %s" % (v, s))

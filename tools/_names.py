import os, sys, types, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import LOSS_CFG, WORKLOADS, synth
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss
from veloxseg_amd.tape_audit import tape_layout
cfg, B = WORKLOADS["autopet128"]
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=2)
x, lab = synth(cfg, B, "cuda", 12345)
eng = TrainEngine(model, crit, (B, 2, 128, 128, 128), use_graph=True, overlap=False)
eng.step(x, lab); torch.cuda.synchronize()
G = eng.graphs
c = collections.Counter()
tot = 0
for key in ("enc_fwd", "loss", "enc_bwd"):
    for n in tape_layout(G[key])[2]: c[(key, n)] += 1
for key in ("dec_fwd", "dec_bwd", "dec_wg"):
    for k, t in enumerate(G[key]):
        for n in tape_layout(t)[2]: c[(key, n)] += 1
for (k, n), v in sorted(c.items()):
    if "wgrad" in n or "wg_" in n or "bias_grad" in n: print(k, v, n)
print("nodes", sum(c.values()))

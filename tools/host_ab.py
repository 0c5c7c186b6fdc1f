"""A/B of host time per eager step in ONE process (alternating), small workload so that the GPU never limits: argv = flag name on veloxseg_amd.functional"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloxseg_amd  # noqa
import torch
from veloxseg_amd import functional as VF
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from recipe import CASES, LOSS_CFG, make_inputs
flag = sys.argv[1] if len(sys.argv) > 1 else "USE_COMPOSITE"
cfg, _ = CASES["g1_48_m2"]
cfg = dict(cfg, proj_drop=0.1, conv_drop=0.1)
x, lab = make_inputs(cfg, 1)
torch.manual_seed(0)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=2)
eng = TrainEngine(model, crit, tuple(x.shape))
eng.step(x.cuda(), lab.cuda())
res = {True: [], False: []}
for rep in range(6):
    for v in (True, False):
        setattr(VF, flag, v)
        for _ in range(3): eng.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): eng.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        res[v].append(((t1 - t0) / 20 * 1e3, (time.perf_counter() - t0) / 20 * 1e3))
for v in (True, False):
    print(flag, v, "host ms/step: min %.2f median %.2f | with sync: min %.2f" % (min(r[0] for r in res[v]), sorted(r[0] for r in res[v])[3], min(r[1] for r in res[v])))

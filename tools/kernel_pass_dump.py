"""Every C-ABI entry of one serialised eager training step with its HIP-event time: argv = substring filter (optional), workload autopet128 B=4"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloxseg_amd  # noqa
import torch
from veloxseg_amd import _hip as H, functional as VF
from bench import WORKLOADS, LOSS_CFG, synth
from veloxseg_amd.engine import TrainEngine
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cfg, B = WORKLOADS["autopet128"]
if os.environ.get("VX_ATTN_DROP"):
    cfg = dict(cfg, attn_drop=float(os.environ["VX_ATTN_DROP"]))
torch.manual_seed(12345)
model = VeloxSeg(**cfg).cuda()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=2)
eng = TrainEngine(model, crit, (B, 2, 128, 128, 128))
x, lab = synth(cfg, B, "cuda", 12345)
for _ in range(3): eng.step(x, lab)
VF.BRANCH_STREAMS = False
eng.flat.reattach(); eng._fwd_bwd_single()
H.profile_begin(); eng._fwd_bwd_single(); prof = H.profile_end()
rows = sorted(((v[1], v[0], k) for k, v in prof.items()), reverse=True)
tot = sum(r[0] for r in rows)
print("total %.3f ms, %d launches" % (tot, sum(r[1] for r in rows)))
for ms, n, (name, key) in rows:
    if flt in name:
        print("%-34s n %3d  total %7.3f ms  each %7.1f us  %s" % (name, n, ms, ms / n * 1e3, list(key)))

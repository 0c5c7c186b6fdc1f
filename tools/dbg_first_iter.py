import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import veloxseg_amd  # noqa
import torch
from recipe import CASES, LOSS_CFG, make_inputs, fill_state_dict
from veloxseg_amd import functional as VF
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss
mode, path = sys.argv[1], sys.argv[2]
cfg, B = CASES["g1_48_m2"]
cfg = dict(cfg, attn_drop=0.0, proj_drop=0.0, conv_drop=0.0)
x, lab = make_inputs(cfg, B)
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=2)
if os.environ.get("NOWS") == "1":
    VF.USE_WGRAD_WS = False
    VF.cpp_module().set_flags(VF.USE_S1, VF.USE_EXPAND_MFMA, VF.USE_GCONV1, False, VF.USE_PATCHIFY, VF.USE_IN_ROW, VF.PW_MFMA_MAX_V, VF.IN_ROW_MAX, VF.IN_EPS, VF.LN_EPS)
    VF._CPP_DEFAULTS = None
if os.environ.get("WARM") == "1":     # touch the allocator with a big block first
    t_ = torch.empty(64 << 20, device="cuda"); del t_
torch.manual_seed(3)
model = VeloxSeg(**cfg).cuda().train()
out = model(x.cuda()); loss = crit(out, lab.cuda(), sr_labels=x.cuda()); loss.backward(); torch.cuda.synchronize()
g = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
if mode == "save":
    torch.save({"loss": float(loss), "g": g, "outs": [o.detach().cpu() for o in out]}, path); print("saved", float(loss))
else:
    ref = torch.load(path)
    print("loss", float(loss), ref["loss"], "max out diff", max(float((a.detach().cpu() - b).abs().max()) for a, b in zip(out, ref["outs"])))
    bad = []
    for n in g:
        d = float((g[n] - ref["g"][n]).norm()); r = float(ref["g"][n].norm())
        if d > 1e-3 * max(r, 1e-4): bad.append((n, round(d, 6), round(r, 6)))
    print(len(bad), "bad params of", len(g)); [print("  ", b) for b in bad[:12]]

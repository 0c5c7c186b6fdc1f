"""Host time of an eager training step by public functional op (count, total us, mean us) -- forward call sites only; plus module-level totals."""
import os, sys, time, types, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloxseg_amd  # noqa
import torch
from veloxseg_amd import functional as VF
from bench import WORKLOADS, LOSS_CFG, synth
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(name):
    f = getattr(VF, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[name][0] += 1; acc[name][1] += time.perf_counter() - t0; return r
    setattr(VF, name, g)
for n in ["conv3d", "conv_transpose_k2s2", "instnorm_sum", "layernorm_cf", "gelu_dropout", "residual_dropout", "add", "space_to_depth2", "pwa_core", "upsample_trilinear", "gram", "veloxseg_loss"]:
    wrap(n)
cfg, _ = WORKLOADS["autopet96"]
torch.manual_seed(0)
model = VeloxSeg(**cfg).cuda().train()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=2)
x, lab = synth(cfg, 1, "cuda", 1)
for p in model.parameters(): VF.grad_buf(p)
def step():
    out = model(x); loss = crit(out, lab, sr_labels=x); loss.backward()
for _ in range(3): step()
torch.cuda.synchronize(); acc.clear()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    tf = time.perf_counter(); out = model(x); loss = crit(out, lab, sr_labels=x); tb = time.perf_counter(); loss.backward(); te = time.perf_counter()
    acc["_forward_total"][0] += 1; acc["_forward_total"][1] += tb - tf; acc["_backward_total"][0] += 1; acc["_backward_total"][1] += te - tb
torch.cuda.synchronize()
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:24s} calls/step {c / N:6.1f}  ms/step {t / N * 1e3:7.3f}  us/call {t / c * 1e6:7.1f}")

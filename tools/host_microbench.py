"""Host-side micro costs (us): one C-ABI call through the fast-call extension vs ctypes, torch.empty, an empty autograd.Function round trip,
and the forward / backward halves of an eager step at a size where the GPU never limits."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloxseg_amd  # noqa
import torch
from veloxseg_amd import _hip as H, functional as VF
from bench import WORKLOADS, LOSS_CFG, synth
from veloxseg_amd.model.VeloxSeg import VeloxSeg
from veloxseg_amd.utils.loss import Loss

def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt

a = torch.zeros(64, device="cuda"); b = torch.zeros(64, device="cuda"); c = torch.zeros(64, device="cuda")
pa, pb, pc = a.data_ptr(), b.data_ptr(), c.data_ptr()
st = H.stream_ptr()
fast = H.fast_module()
dll = H.LIB.load()
print("fast vx_add            %.2f us" % t(lambda: fast.vx_add(pa, pb, None, pc, 64, st)))
print("ctypes vx_add          %.2f us" % t(lambda: dll.vx_add(pa, pb, None, pc, 64, st)))
print("H.call vx_add (+P x3)  %.2f us" % t(lambda: H.call("vx_add", H.P(a), H.P(b), None, H.P(c), 64, H.stream_ptr())))
print("aten add_              %.2f us" % t(lambda: a.add_(b)))
print("torch.empty            %.2f us" % t(lambda: torch.empty((4, 16, 8, 8, 8), device="cuda")))
print("stream_ptr             %.2f us" % t(lambda: H.stream_ptr()))
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x); return x.view_as(x)
    @staticmethod
    def backward(ctx, g):
        return g
xr = torch.zeros(8, device="cuda", requires_grad=True)
print("Function.apply fwd     %.2f us" % t(lambda: F.apply(xr)))
y = VF.add(a, b)
print("VF.add (no grad)       %.2f us" % t(lambda: VF.add(a, b)))
ar = a.clone().requires_grad_(True)
print("VF.add (grad)          %.2f us" % t(lambda: VF.add(ar, b)))
cfg, _ = WORKLOADS["autopet96"]
torch.manual_seed(0)
model = VeloxSeg(**cfg).cuda().train()
crit = Loss(types.SimpleNamespace(model_name="VeloxSeg"), LOSS_CFG, None, num_modal=2)
x, lab = synth(cfg, 1, "cuda", 1)
for p in model.parameters(): VF.grad_buf(p)
def step():
    t0 = time.perf_counter(); out = model(x); t1 = time.perf_counter(); loss = crit(out, lab, sr_labels=x); t2 = time.perf_counter(); loss.backward(); t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2
for _ in range(3): step()
torch.cuda.synchronize()
r = [step() for _ in range(10)]
print("forward %.2f ms, loss %.2f ms, backward %.2f ms (host, no sync)" % tuple(sum(v[i] for v in r) / len(r) * 1e3 for i in range(3)))
with torch.no_grad():
    model.eval()
    for _ in range(3): model(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): model(x)
    print("eval forward (no grad) %.2f ms host" % ((time.perf_counter() - t0) * 100))

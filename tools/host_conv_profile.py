import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloxseg_amd  # noqa
import torch
from veloxseg_amd import functional as VF, _hip as H
x = torch.randn(4, 32, 8, 8, 8, device="cuda", requires_grad=True)
w = torch.randn(32, 32, 1, 1, 1, device="cuda", requires_grad=True)
b = torch.zeros(32, device="cuda", requires_grad=True)
VF.grad_buf(w); VF.grad_buf(b)
def f():
    return VF.conv3d(x, w, b)
for _ in range(100): f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): f()
print("conv3d fwd (pw mfma) %.2f us/call" % ((time.perf_counter() - t0) / 2000 * 1e6))
torch.cuda.synchronize()
g = torch.ones(4, 32, 8, 8, 8, device="cuda")
ys = [f() for _ in range(500)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for y in ys: y.backward(g)
print("conv3d bwd           %.2f us/call" % ((time.perf_counter() - t0) / 500 * 1e6))
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): f()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

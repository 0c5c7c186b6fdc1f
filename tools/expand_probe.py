#!/usr/bin/env python3
"""Time the patch-expand MFMA entries (forward, input gradient) with their operands in LDS or loaded per tap, at the autopet128 B=4 shapes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from veloxseg_amd import _hip as H
d = torch.device("cuda:0")
B, S = 4, 32
st = H.stream_ptr()
def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for Cc in (2, 1):
    Cout = 64 * Cc
    x = torch.randn(B, 16, S, S, S, device=d); w = torch.randn(Cout, 16, 3, 3, 3, device=d) * 0.1; bias = torch.randn(Cout, device=d)
    wt = torch.empty(Cout * 16 * 27, device=d); y = torch.empty(B, Cc, 4 * S, 4 * S, 4 * S, device=d); dx = torch.empty_like(x)
    fl = 2.0 * B * S ** 3 * Cout * 16 * 27
    for knob, vals in (("vx_expand_set_fwd_wlds", (0, 1)),):
        for v in vals:
            H.call(knob, v)
            t = timeit(lambda: H.call("vx_expand_fwd_mfma", H.P(x), H.P(w), H.P(bias), H.P(wt), H.P(y), B, Cc, S, S, S, st))
            print(f"Cc={Cc} forward  {knob}={v}: {t:7.1f} us  {fl / t / 1e6:5.1f} TFLOP/s")
        H.call(knob, 1)
    for v in (2, 1):
        H.call("vx_expand_set_lds", v)
        t = timeit(lambda: H.call("vx_expand_bwd_data_mfma", H.P(y), H.P(w), H.P(wt), H.P(dx), B, Cc, S, S, S, 0, st))
        print(f"Cc={Cc} backward vx_expand_set_lds={v}: {t:7.1f} us  {fl / t / 1e6:5.1f} TFLOP/s")
    H.call("vx_expand_set_lds", 1)

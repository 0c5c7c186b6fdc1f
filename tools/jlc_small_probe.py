#!/usr/bin/env python3
"""How fast are the fused JLC spatial kernels (csrc/jlc.hip) at the 8^3 / 4^3 levels, where the block still runs per-operator kernels?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from veloxseg_amd import _hip as H
d = torch.device("cuda:0")
for (B, C, G, S) in ((4, 64, 8, 8), (4, 128, 8, 4), (4, 32, 4, 16)):
    V = S ** 3
    x = torch.randn(B, C, S, S, S, device=d)
    cg = C // G
    w1, w3, w5 = (torch.randn(C, cg, k, k, k, device=d) * 0.05 for k in (1, 3, 5))
    b1, b3, b5 = (torch.zeros(C, device=d) for _ in range(3))
    y = torch.empty(3, B, C, S, S, S, device=d)
    nty = H.query("vx_jlc_ntiles", B, C, G, S, S, S)
    nch = H.query("vx_jlc_nchunks", B * C, V)
    part_y = torch.empty(3, B * C, nty, 2, device=d, dtype=torch.float64)
    part_o = torch.empty(B * C, nch, 2, device=d, dtype=torch.float64)
    stats_y = torch.empty(3, B * C, 2, device=d)
    o = torch.empty_like(x)
    n1 = B * C * V
    yp = y.data_ptr()
    st = H.stream_ptr()

    def fwd():
        H.call("vx_jlc_conv_fwd", H.P(x), H.P(w1), H.P(w3), H.P(w5), H.P(b1), H.P(b3), H.P(b5), yp, yp + 4 * n1, yp + 8 * n1, H.P(part_y, torch.float64), B, C, G, S, S, S, st)
        H.call("vx_jlc_mid_fwd", H.P(x), yp, yp + 4 * n1, yp + 8 * n1, H.P(part_y, torch.float64), nty, H.P(stats_y), H.P(o), H.P(part_o, torch.float64), B * C, V, 1e-5, st)
    g = torch.randn(3, B, C, S, S, S, device=d)
    dx = torch.empty_like(x)
    gp = g.data_ptr()

    def bwd():
        H.call("vx_jlc_conv_bwd", gp, gp + 4 * n1, gp + 8 * n1, H.P(w1), H.P(w3), H.P(w5), H.P(x), H.P(dx), B, C, G, S, S, S, st)
    for name, fn in (("conv_fwd + mid_fwd", fwd), ("conv_bwd", bwd)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"B{B} C{C} G{G} {S}^3  {name:20s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us  (tiles {nty})")

#!/usr/bin/env python3
"""vx_jlc_wgrad_tz at the three JLC levels of the 128^3 configurations: time of the production kernel, and of the timing-experiment build with phases switched off
(vx_jlc_tz_set_debug, weight-gradient bits << 4: 1 no staging, 2 no MFMA phase, 4 no fold / atomics, 8 no MFMAs, 16 no A reads)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from veloxseg_amd import _hip as H
d = torch.device("cuda:0")
st = H.stream_ptr()
for (B, C, G, S) in ((4, 16, 4, 32), (4, 32, 4, 16), (4, 64, 8, 8)):
    x = torch.randn(B, C, S, S, S, device=d)
    g = torch.randn(3, B, C, S, S, S, device=d)
    cg = C // G
    dws = [torch.zeros(C, cg, k, k, k, device=d) for k in (1, 3, 5)]
    n1 = B * C * S ** 3
    gp = g.data_ptr()

    def run():
        H.call("vx_jlc_wgrad_tz_ns", H.P(x), gp, gp + 4 * n1, gp + 8 * n1, H.P(dws[0]), H.P(dws[1]), H.P(dws[2]), B, C, G, S, S, S, 22, st)
    row = []
    for dbg in (0, 1, 2, 4, 1 | 2, 1 | 2 | 4, 8, 16):
        H.call("vx_jlc_tz_set_debug", dbg << 4)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record(); torch.cuda.synchronize()
        row.append(f"dbg {dbg}: {e0.elapsed_time(e1) / 50 * 1e3:5.1f}")
    H.call("vx_jlc_tz_set_debug", 0)
    print(f"B{B} C{C} G{G} {S}^3   " + "  ".join(row) + "  us")

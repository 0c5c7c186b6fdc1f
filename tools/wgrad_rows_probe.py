#!/usr/bin/env python3
"""Time the JLC weight-gradient kernels (row-sliding vs (ci, tap)-pair kernel) at the shapes of the autopet128 B=4 step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from veloxseg_amd import _hip as H
d = torch.device("cuda:0")
B = 4
for (C, G, K, S) in [(16, 4, 5, 32), (16, 4, 3, 32), (32, 4, 5, 16), (32, 4, 3, 16), (64, 8, 5, 8), (64, 8, 3, 8), (128, 16, 5, 4), (128, 16, 3, 4), (16, 4, 5, 24), (32, 4, 5, 12)]:
    x = torch.randn(B, C, S, S, S, device=d)
    dy = torch.randn(B, C, S, S, S, device=d)
    dw = torch.zeros(C, C // G, K, K, K, device=d)
    st = H.stream_ptr()
    res = []
    for rows in (0, 1):
        H.call("vx_wgrad_set_rows", rows)
        for _ in range(3):
            H.call("vx_conv3d_bwd_weight_tiled", H.P(x), None, C, H.P(dy), H.P(dw), None, B, C, S, S, S, C, K, 1, K // 2, G, 1, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            H.call("vx_conv3d_bwd_weight_tiled", H.P(x), None, C, H.P(dy), H.P(dw), None, B, C, S, S, S, C, K, 1, K // 2, G, 1, st)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    H.call("vx_wgrad_set_rows", 1)
    fl = 2.0 * B * S ** 3 * C * (C // G) * K ** 3
    print(f"C={C:3d} Cg={C // G} K={K} {S}^3: pairs {res[0]:7.1f} us ({fl / res[0] / 1e6:5.1f} TFLOP/s)   rows {res[1]:7.1f} us ({fl / res[1] / 1e6:5.1f} TFLOP/s)")

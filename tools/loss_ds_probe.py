#!/usr/bin/env python3
"""Time the fused deep-supervision loss kernels (csrc/loss_ds.hip) at the autopet128 B=4 shape."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from veloxseg_amd import functional as VF
d = torch.device("cuda:0")
B, C, S = 4, 2, 128
heads = [torch.randn(B, C, S, S, S, device=d, requires_grad=True)] + [torch.randn(B, C, s, s, s, device=d, requires_grad=True) for s in (16, 8, 4)]
lab = (torch.rand(B, 1, S, S, S, device=d) > 0.97).to(torch.uint8 if os.environ.get("VX_LAB", "u8") == "u8" else torch.int64)
w = (0.25,) * 4
tf = tb = 0.0
for it in range(8):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    loss = VF.seg_only_loss(heads, lab, w)
    e[1].record()
    loss.backward()
    e[2].record()
    torch.cuda.synchronize()
    if it >= 3:
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
print(f"dbg={os.environ.get('VX_DS_DBG', '0')}: fwd {tf / 5 * 1e3:.1f} us  bwd {tb / 5 * 1e3:.1f} us")

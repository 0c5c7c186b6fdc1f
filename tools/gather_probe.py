#!/usr/bin/env python3
"""Time vx_pwa_gather_all_fwd / _bwd alone at the headline network's first PWA level (32^3 grid, 4^3 windows, four scales, c_qk = c_v = 4, M = 2, B = 2),
all scales together and one scale at a time: microseconds per launch and the bytes each one has to move."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from veloxseg_amd import _hip as H

B, M, heads, cq, cv, G, n = 2, 2, 1, 4, 4, 32, 4
dev = torch.device("cuda")


def run(scales, reps=50):
    small = [[s] * 3 for s in scales]
    nwin = [[G // (n * s)] * 3 for s in scales]
    plan = H.make_plan([G] * 3, [n] * 3, heads, small, nwin)
    nb = len(scales)
    V = G ** 3
    src = [torch.randn(B, nb * heads * (cv if k % 3 == 2 else cq), G, G, G, device=dev) for k in range(3 * M)]
    ML = M * plan.l
    tq = torch.empty(B, heads, plan.Ntot, ML, cq, device=dev); tk = torch.empty_like(tq); tv = torch.empty(B, heads, plan.Ntot, ML, cv, device=dev)
    iq = torch.zeros(tq.shape, device=dev, dtype=torch.int32); ik = torch.zeros_like(iq); iv = torch.zeros(tv.shape, device=dev, dtype=torch.int32)
    arr = (ctypes.c_void_p * (3 * M))(*[H.P(t) for t in src])
    dst = [torch.empty_like(t) for t in src]
    darr = (ctypes.c_void_p * (3 * M))(*[H.P(t) for t in dst])
    pp = ctypes.addressof(plan)
    st = H.stream_ptr()
    out = []
    for name, fn in (("fwd", lambda: H.call("vx_pwa_gather_all_fwd", ctypes.addressof(arr), H.P(tq), H.P(tk), H.P(tv), H.P(iq, torch.int32), H.P(ik, torch.int32), H.P(iv, torch.int32), pp, cq, cv, M, B, st)),
                     ("bwd", lambda: H.call("vx_pwa_gather_all_bwd", H.P(tq), H.P(tk), H.P(tv), H.P(iq, torch.int32), H.P(ik, torch.int32), H.P(iv, torch.int32), ctypes.addressof(darr), pp, cq, cv, M, B, st))):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append((name, e0.elapsed_time(e1) * 1000 / reps))
    vol = sum(t.numel() for t in src) * 4
    tok = (tq.numel() * 2 + tv.numel()) * 4
    print(f"scales {scales}: volume {vol / 1e6:.1f} MB, tokens {tok / 1e6:.1f} MB | " + " | ".join(f"{k} {v:.1f} us ({(vol + tok) / v / 1e3:.0f} GB/s)" for k, v in out), flush=True)


for sc in ([1, 2, 4, 8], [1], [2], [4], [8]):
    run(sc)

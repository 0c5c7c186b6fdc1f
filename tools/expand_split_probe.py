#!/usr/bin/env python3
"""Patch-expand layer (Conv3d 16 -> 64 k + PixelShuffle(4), Decoder.py:73-76,150-153) at the bench shapes: fp32 MFMA vs split-bf16 with 2 / 3 pieces (3 / 6 bf16
MFMAs per pair) vs plain bf16 -- microseconds per launch (HIP events) and the error against an fp64 CPU convolution on a slice."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from veloxseg_amd import _hip as H

d = torch.device("cuda")
for (B, Cc, S) in [(4, 2, 32), (4, 1, 32), (4, 2, 24)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, 16, S, S, S, device=d, generator=g)
    w = torch.randn(64 * Cc, 16, 3, 3, 3, device=d, generator=g) * (2.0 / 432) ** 0.5
    b = torch.randn(64 * Cc, device=d, generator=g) * 0.1
    y = torch.empty(B, Cc, 4 * S, 4 * S, 4 * S, device=d)
    dy = torch.randn(B, Cc, 4 * S, 4 * S, 4 * S, device=d, generator=g)
    dx = torch.empty_like(x)
    ws = torch.empty(max(64 * Cc * 16 * 27, H.query("vx_expand_split_ws_floats", Cc, 3)), device=d)
    st = H.stream_ptr()
    # fp64 reference on sample 0, a 12-row slab
    xs = x[:1].double().cpu()
    ref = F.conv3d(xs, w.double().cpu(), b.double().cpu(), padding=1)                     # (1, 64 Cc, S, S, S)
    refy = ref.view(1, Cc, 4, 4, 4, S, S, S).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(1, Cc, 4 * S, 4 * S, 4 * S)
    gref = torch.autograd.functional.vjp(lambda t: F.conv3d(t, w.double().cpu(), None, padding=1), xs,
                                         dy[:1].double().cpu().view(1, Cc, S, 4, S, 4, S, 4).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(1, 64 * Cc, S, S, S))[1]

    def timeit(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    rows = []
    for name, fwd, bwd in [
        ("fp32 MFMA", lambda: H.call("vx_expand_fwd_mfma", H.P(x), H.P(w), H.P(b), H.P(ws), H.P(y), B, Cc, S, S, S, st),
         lambda: H.call("vx_expand_bwd_data_mfma", H.P(dy), H.P(w), H.P(ws), H.P(dx), B, Cc, S, S, S, 0, st)),
        ("split 3 pieces", lambda: H.call("vx_expand_fwd_mfma_split", H.P(x), H.P(w), H.P(b), H.P(ws), H.P(y), B, Cc, S, S, S, 3, st),
         lambda: H.call("vx_expand_bwd_data_mfma_split", H.P(dy), H.P(w), H.P(ws), H.P(dx), B, Cc, S, S, S, 0, 3, st)),
        ("split 2 pieces", lambda: H.call("vx_expand_fwd_mfma_split", H.P(x), H.P(w), H.P(b), H.P(ws), H.P(y), B, Cc, S, S, S, 2, st),
         lambda: H.call("vx_expand_bwd_data_mfma_split", H.P(dy), H.P(w), H.P(ws), H.P(dx), B, Cc, S, S, S, 0, 2, st)),
        ("bf16", lambda: H.call("vx_expand_fwd_mfma_bf16", H.P(x), H.P(w), H.P(b), H.P(ws), H.P(y), B, Cc, S, S, S, st),
         lambda: H.call("vx_expand_bwd_data_mfma_bf16", H.P(dy), H.P(w), H.P(ws), H.P(dx), B, Cc, S, S, S, 0, st))]:
        tf, tb = timeit(fwd), timeit(bwd)
        fwd(); bwd(); torch.cuda.synchronize()
        ef = float((y[:1].double().cpu() - refy).abs().max()) / float(refy.abs().max())
        eb = float((dx[:1].double().cpu() - gref).abs().max()) / float(gref.abs().max())
        gf = 2.0 * B * S ** 3 * 64 * Cc * 16 * 27 / 1e9
        rows.append(f"  {name:16s} fwd {tf:7.1f} us ({gf / tf * 1e3:6.1f} TFLOP/s)  max err / max |y| {ef:.2e}   dX {tb:7.1f} us ({gf / tb * 1e3:6.1f} TFLOP/s)  err {eb:.2e}")
    print(f"B={B} Cc={Cc} grid {S}^3:")
    print("\n".join(rows), flush=True)
    # ---- weight gradient: dW / db against an fp64 reference (all samples, computed on the GPU in double) ----
    dyc = dy.view(B, Cc, S, 4, S, 4, S, 4).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(B, 64 * Cc, S, S, S)           # coarse view (co = (c, s1, s2, s3))
    wref = torch.nn.grad.conv3d_weight(x.double(), w.shape, dyc.double(), padding=1).cpu()
    bref = dyc.double().sum((0, 2, 3, 4)).cpu()
    xcl = torch.empty(B * S ** 3 * 16, device=d)
    nws = H.query("vx_expand_wgrad_split_ws_floats", B, Cc, S, S, S)
    pws = torch.empty(nws, device=d)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    rows = []
    for name, fn in [("fp32 MFMA", lambda: H.call("vx_expand_wgrad_mfma", H.P(x), H.P(xcl), H.P(dy), H.P(dw), H.P(db), B, Cc, S, S, S, st)),
                     ("split 3 pieces", lambda: H.call("vx_expand_wgrad_mfma_split", H.P(x), H.P(dy), H.P(dw), H.P(db), H.P(pws), nws, B, Cc, S, S, S, 3, st)),
                     ("split 2 pieces", lambda: H.call("vx_expand_wgrad_mfma_split", H.P(x), H.P(dy), H.P(dw), H.P(db), H.P(pws), nws, B, Cc, S, S, S, 2, st))]:
        t = timeit(fn)
        dw.zero_(); db.zero_()
        fn(); torch.cuda.synchronize()
        ew = float((dw.double().cpu() - wref).abs().max()) / float(wref.abs().max())
        eb = float((db.double().cpu() - bref).abs().max()) / float(bref.abs().max())
        gf = 2.0 * B * S ** 3 * 64 * Cc * 16 * 27 / 1e9
        rows.append(f"  {name:16s} dW {t:7.1f} us ({gf / t * 1e3:6.1f} TFLOP/s)  max err / max |dW| {ew:.2e}   db err {eb:.2e}")
    print("\n".join(rows), flush=True)

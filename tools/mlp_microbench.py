#!/usr/bin/env python3
"""Time vx_mlp_fwd / vx_mlp_bwd (csrc/mlp.hip) alone at the bench shapes; at the bench shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import veloxseg_amd  # noqa
import torch
from veloxseg_amd import _hip as H
from veloxseg_amd import functional as VF

dev = torch.device("cuda")
def run(C, R, V, norm, p, B=4, reps=20):
    x = torch.randn(B, C, V, device=dev); dout = torch.randn_like(x)
    w1 = torch.randn(R, C, device=dev) * 0.2; b1 = torch.randn(R, device=dev) * 0.1; w2 = torch.randn(C, R, device=dev) * 0.2; b2 = torch.randn(C, device=dev) * 0.1
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    stats = torch.stack([x.mean(2).flatten(), 1.0 / (x.var(2, unbiased=False).flatten() + 1e-5).sqrt()], 1).contiguous()
    out = torch.empty_like(x); dx = torch.empty_like(x)
    npd = H.query("vx_mlp_bwd_nparts", B, C, V)
    part = torch.empty(B * C * npd * 2, device=dev)
    g = [torch.zeros_like(t) for t in (gamma, beta, w1, b1, w2, b2)]
    rs = VF.rng_state(dev)
    st = H.stream_ptr()
    def fwd():
        H.call("vx_mlp_fwd", H.P(x), norm, None, 0, H.P(stats), H.P(gamma), H.P(beta), H.P(w1), H.P(b1), H.P(w2), H.P(b2), H.P(out), B, C, R, V, 1e-5, rs.data_ptr(), 1, p if norm else 0.0, 2, p, st)
    def bwd():
        H.call("vx_mlp_bwd", H.P(x), norm, H.P(stats), H.P(gamma), H.P(beta), H.P(w1), H.P(b1), H.P(w2), H.P(dout), H.P(dx), H.P(part), H.P(g[0]), H.P(g[1]), H.P(g[2]), H.P(g[3]), H.P(g[4]), H.P(g[5]),
               B, C, R, V, 1e-5, rs.data_ptr(), 1, p if norm else 0.0, 2, p, st)
    res = []
    for f in (fwd, bwd):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / reps * 1e3)
    print(f"C={C} R={R} V={V} norm={norm} p={p} dbg={os.environ.get('VX_MLP_DBG', '0')}: fwd {res[0]:.1f} us  bwd {res[1]:.1f} us", flush=True)

for C, R, V in ((16, 48, 32768), (32, 96, 4096)):
    for norm, p in ((0, 0.0), (0, 0.1), (1, 0.1)):
        run(C, R, V, norm, p)

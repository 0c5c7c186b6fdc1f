#!/usr/bin/env python3
"""Attention backward alone at the bench shapes (autopet128 / autopet96, B = 4, dropout 0.1): one-pass MFMA kernel (mask 3) vs the two-kernel MFMA
backward (mask 5, where its geometry allows) vs the VALU kernels (mask 0); microseconds per call (HIP events, 20 calls after 3 warm-up calls)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from veloxseg_amd import _hip as H
from veloxseg_amd import functional as VF
from oracle import veloxseg_oracle as O          # plan geometry only (tools are not the product)

ONLY = os.environ.get("VX_PROBE_ONLY")
LEVELS = {"128": [([32] * 3, [4] * 3, 1, 4, 16), ([16] * 3, [8] * 3, 2, 8, 32), ([8] * 3, [4] * 3, 2, 8, 64), ([4] * 3, [4] * 3, 4, 16, 128)],
          "96": [([24] * 3, [3] * 3, 1, 4, 16), ([12] * 3, [6] * 3, 2, 8, 32), ([6] * 3, [3] * 3, 2, 8, 64), ([3] * 3, [3] * 3, 4, 16, 128)],
          # config/models_config_hecktor2022.json: 128 x 128 x 64 patch, anisotropic windows (l = 32 / 256 tokens)
          "heck": [([32, 32, 16], [4, 4, 2], 1, 4, 16), ([16, 16, 8], [8, 8, 4], 2, 8, 32), ([8, 8, 4], [4, 4, 2], 2, 8, 64), ([4, 4, 2], [4, 4, 2], 4, 16, 128)]}
B, M, p = int(os.environ.get("VX_PROBE_B", "4")), int(os.environ.get("VX_PROBE_M", "2")), 0.1
d = torch.device("cuda")
for name, levels in LEVELS.items():
    for L, (grid, big, heads, mdh, C) in enumerate(levels, 1):
        if ONLY and f"{name}L{L}" not in ONLY.split(","):
            continue
        pl = O.plan_pwa(grid, big, [1, 1, 1], 2, heads, mdh, C)
        plan = H.make_plan(grid, pl["n"], heads, pl["small"], pl["nwin"])
        pp = H.ctypes.addressof(plan)
        cq, cv, Nt, ML = pl["c_qk"], pl["c_v"], plan.Ntot, M * plan.l
        g = torch.Generator(device="cuda").manual_seed(1)
        tq, tk = (torch.randn(B, heads, Nt, ML, cq, device=d, generator=g) for _ in range(2))
        tv, dO = (torch.randn(B, heads, Nt, ML, cv, device=d, generator=g) for _ in range(2))
        n = pl["n"]
        table = 0.5 * torch.randn((2 * n[0] - 1) * (2 * n[1] - 1) * (2 * n[2] - 1), heads, device=d, generator=g)
        Oo, lse = torch.empty_like(tv), torch.empty(B, heads, Nt, ML, device=d)
        rs = VF.rng_state(d)
        st = H.stream_ptr()
        mbits = torch.empty(H.query("vx_pwa_attn_mbits_words", pp, B, M), device=d, dtype=torch.int16)
        H.call("vx_pwa_attn_fwd_mb", H.P(tq), H.P(tk), H.P(tv), H.P(table), H.P(Oo), H.P(lse), pp, B, M, cq, cv, H.P(rs, torch.int64), 5, p, H.P(mbits, torch.int16), st)
        nws = H.query("vx_pwa_attn_bwd_ws_floats", pp, B, M)
        ws = torch.empty(nws, device=d)
        dq, dk, dv, dt = torch.empty_like(tq), torch.empty_like(tk), torch.empty_like(tv), torch.zeros_like(table)
        # the f16-pipe one-pass kernel (levels 1 / 2 of 128^3) against the fp32 VALU kernels: time and the largest difference of dq / dk / dv / d(table)
        if H.query("vx_pwa_attn_bwd1h_ok", pp, B, M, cq, cv) == 1:
            res = {}
            for f16 in (1, 0):
                H.call("vx_pwa_attn_set_f16_bwd", 1 if f16 else 0)      # (VELOXSEG_F16_BWD_QS in the environment: A/B of the query splits per key chunk)
                dt.zero_()

                def run():
                    H.call("vx_pwa_attn_bwd_mb", H.P(tq), H.P(tk), H.P(tv), H.P(table), H.P(Oo), H.P(lse), H.P(dO), H.P(dq), H.P(dk), H.P(dv), H.P(dt), H.P(ws), pp, B, M, cq, cv,
                           H.P(rs, torch.int64), 5, p, H.P(mbits, torch.int16) if f16 else None, st)
                run()
                torch.cuda.synchronize()
                res[f16] = [x.clone() for x in (dq, dk, dv, dt)]
                for _ in range(3):
                    run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run()
                e1.record()
                torch.cuda.synchronize()
                res[f16].append(e0.elapsed_time(e1) / 20 * 1e3)
            err = [float((a - b).abs().max()) / max(1e-30, float(b.abs().max())) for a, b in zip(res[1][:4], res[0][:4])]
            print(f"      f16-pipe one-pass backward {res[1][4]:7.1f} us  vs fp32 kernels {res[0][4]:7.1f} us   max |diff| / max: dq {err[0]:.1e} dk {err[1]:.1e} dv {err[2]:.1e} dtable {err[3]:.1e}", flush=True)
            H.call("vx_pwa_attn_set_f16_bwd", 0)
        out = []
        for mask in (11, 11.5, 5, 0):            # 11 = one pass (forced for every geometry) with the forward's mask bits, 11.5 = one pass drawing the Philox words again
            H.call("vx_pwa_attn_set_mfma", int(mask))
            if mask == 5 and not (H.query("vx_pwa_attn_mfma_ok", pp, B, M, cq, cv) & 2):
                out.append("   -  ")
                continue

            def run():
                H.call("vx_pwa_attn_bwd_mb", H.P(tq), H.P(tk), H.P(tv), H.P(table), H.P(Oo), H.P(lse), H.P(dO), H.P(dq), H.P(dk), H.P(dv), H.P(dt), H.P(ws), pp, B, M, cq, cv,
                       H.P(rs, torch.int64), 5, p, H.P(mbits, torch.int16) if mask == 11 else None, st)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            out.append(f"{e0.elapsed_time(e1) / 20 * 1e3:7.1f}")
        H.call("vx_pwa_attn_set_mfma", 3)
        # the VALU kernels reading the forward's keep bits (aligned windows) vs re-drawing the Philox words
        vb = []
        for use in (True, False):
            def run2():
                H.call("vx_pwa_attn_bwd_mb", H.P(tq), H.P(tk), H.P(tv), H.P(table), H.P(Oo), H.P(lse), H.P(dO), H.P(dq), H.P(dk), H.P(dv), H.P(dt), H.P(ws), pp, B, M, cq, cv,
                       H.P(rs, torch.int64), 5, p, H.P(mbits, torch.int16) if use else None, st)
            H.call("vx_pwa_attn_set_mfma", 0)
            for _ in range(3):
                run2()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run2()
            e1.record()
            torch.cuda.synchronize()
            vb.append(e0.elapsed_time(e1) / 20 * 1e3)
        H.call("vx_pwa_attn_set_mfma", 3)
        print(f"      VALU backward with the forward's keep bits {vb[0]:7.1f} us, re-drawing {vb[1]:7.1f} us")
        H.call("vx_pwa_attn_set_f16_bwd", 1)
        pairs = B * heads * Nt * ML * ML
        print(f"{name}^3 L{L}: l={plan.l:4d} ML={ML:5d} windows={B * heads * Nt:5d} c_qk/c_v={cq}/{cv}  pairs={pairs / 1e6:7.1f}M   one-pass {out[0]} us (Philox again: {out[1]}) | two-kernel MFMA {out[2]} us | VALU {out[3]} us", flush=True)

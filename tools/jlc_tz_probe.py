"""A/B of the JLC grouped convolutions: fp32 VALU kernels (csrc/jlc.hip) vs the Toeplitz-MFMA kernels (csrc/jlc_mfma.hip), forward and input gradient, at the
shapes of the four encoder levels.  Prints the error of each against an fp64 torch convolution and the stand-alone launch times (HIP events, 50 launches).

    python tools/jlc_tz_probe.py [--pieces 3] [--batch 4] [--size 128|96]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veloxseg_amd import _hip as H  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pieces", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--levels", default="1,2,3,4")
    ap.add_argument("--breakdown", action="store_true")
    ap.add_argument("--cl", action="store_true", help="levels the channels-last kernels cover (<= 8 voxels per axis) run those instead of the Toeplitz kernels")
    args = ap.parse_args()
    H.LIB.load()
    H.call("vx_jlc_tz_set_pieces", args.pieces)
    H.call("vx_jlc_tz_set_min_voxels", 0)          # (the probe measures every level; the library selects the matrix-pipe convolutions from 16^3 up)
    B = args.batch
    g1 = args.size // 4
    levels = {1: (16, 4, g1), 2: (32, 4, g1 // 2), 3: (64, 8, g1 // 4), 4: (128, 8, g1 // 8)}
    st = torch.cuda.current_stream().cuda_stream
    for L in [int(v) for v in args.levels.split(",")]:
        C, G, S = levels[L]
        D = Hh = W = S
        ok = H.query("vx_jlc_tz_ok", C, G, D, Hh, W)
        use_cl = args.cl and H.query("vx_jlc_cl_ok", C, G, D, Hh, W) == 1          # the coarse levels: channels-last kernels of csrc/jlc_cl.hip instead
        pre = "vx_jlc_cl" if use_cl else "vx_jlc_tz"
        print(f"--- level {L}: B={B} C={C} G={G} {D}x{Hh}x{W}  tz_ok={ok}  kernels: {pre}")
        if not ok and not use_cl:
            continue
        torch.manual_seed(L)
        Cg = C // G
        x = torch.randn(B, C, D, Hh, W, device="cuda")
        ws = [torch.randn(C, Cg, k, k, k, device="cuda") * (1.0 / (Cg * k ** 3) ** 0.5) for k in (1, 3, 5)]
        bs = [torch.randn(C, device="cuda") * 0.1 for _ in range(3)]
        n1 = x.numel()
        # ---------------- forward
        y_old, y_new = torch.empty(3, *x.shape, device="cuda"), torch.empty(3, *x.shape, device="cuda")
        nt_old = H.query("vx_jlc_ntiles", B, C, G, D, Hh, W)
        nt_new = H.query(pre + "_ntiles", C, G, D, Hh, W)
        p_old = torch.empty(3, B * C, nt_old, 2, device="cuda", dtype=torch.float64)
        p_new = torch.empty(3, B * C, nt_new, 2, device="cuda", dtype=torch.float64)
        img = torch.empty(H.query(pre + "_img_floats", C, G), device="cuda")

        def f_old():
            H.call("vx_jlc_conv_fwd", H.P(x), H.P(ws[0]), H.P(ws[1]), H.P(ws[2]), H.P(bs[0]), H.P(bs[1]), H.P(bs[2]), y_old[0].data_ptr(), y_old[1].data_ptr(),
                   y_old[2].data_ptr(), p_old.data_ptr(), B, C, G, D, Hh, W, st)

        def f_prep():
            H.call(pre + "_prep", H.P(ws[0]), H.P(ws[1]), H.P(ws[2]), H.P(img), C, G, st)

        def f_new():
            H.call(pre + "_fwd", H.P(x), H.P(img), H.P(bs[0]), H.P(bs[1]), H.P(bs[2]), y_new[0].data_ptr(), y_new[1].data_ptr(), y_new[2].data_ptr(),
                   p_new.data_ptr(), B, C, G, D, Hh, W, st)

        f_old(); f_prep(); f_new()
        torch.cuda.synchronize()
        xd = x.double()
        for i, k in enumerate((1, 3, 5)):
            ref = TF.conv3d(xd, ws[i].double(), bs[i].double(), padding=k // 2, groups=G)
            sc = float(ref.abs().max())
            eo, en = float((y_old[i].double() - ref).abs().max()) / sc, float((y_new[i].double() - ref).abs().max()) / sc
            so = float((p_old[i].sum(1)[:, 0].view(B, C) - ref.sum((2, 3, 4))).abs().max()), float((p_old[i].sum(1)[:, 1].view(B, C) - (ref * ref).sum((2, 3, 4))).abs().max())
            sn = float((p_new[i].sum(1)[:, 0].view(B, C) - ref.sum((2, 3, 4))).abs().max()), float((p_new[i].sum(1)[:, 1].view(B, C) - (ref * ref).sum((2, 3, 4))).abs().max())
            print(f"  fwd k={k}: rel-to-max err  valu {eo:.2e}  tz {en:.2e}   stats err valu {so[0]:.1e}/{so[1]:.1e}  tz {sn[0]:.1e}/{sn[1]:.1e}")
        t_old, t_prep, t_new = timeit(f_old), timeit(f_prep), timeit(f_new)
        flops = 2.0 * B * C * D * Hh * W * Cg * 153
        print(f"  fwd time: valu {t_old:.1f} us ({flops / t_old / 1e6:.1f} TFLOP/s)  tz {t_new:.1f} us ({flops / t_new / 1e6:.1f} TFLOP/s)  prep {t_prep:.1f} us   tiles {nt_old} / {nt_new}")
        # ---------------- input gradient
        g = torch.randn(3, *x.shape, device="cuda")
        d_o = torch.randn_like(x)
        dx_old, dx_new = torch.empty_like(x), torch.empty_like(x)

        def b_old():
            H.call("vx_jlc_conv_bwd", g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), H.P(ws[0]), H.P(ws[1]), H.P(ws[2]), H.P(d_o), H.P(dx_old), B, C, G, D, Hh, W, st)

        def b_new():
            if use_cl:
                H.call("vx_jlc_cl_bwd", g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), H.P(img), H.P(d_o), H.P(dx_new), B, C, G, D, Hh, W, st)
            else:
                H.call("vx_jlc_tz_bwd", g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), H.P(img), H.P(ws[0]), H.P(d_o), H.P(dx_new), B, C, G, D, Hh, W, st)

        b_old(); b_new()
        torch.cuda.synchronize()
        ref = d_o.double()
        for i, k in enumerate((1, 3, 5)):
            ref = ref + TF.conv_transpose3d(g[i].double(), ws[i].double(), None, padding=k // 2, groups=G)
        sc = float(ref.abs().max())
        print(f"  bwd: rel-to-max err  valu {float((dx_old.double() - ref).abs().max()) / sc:.2e}  tz {float((dx_new.double() - ref).abs().max()) / sc:.2e}")
        t_old, t_new = timeit(b_old), timeit(b_new)
        print(f"  bwd time: valu {t_old:.1f} us ({flops / t_old / 1e6:.1f} TFLOP/s)  tz {t_new:.1f} us ({flops / t_new / 1e6:.1f} TFLOP/s)")
        # ---------------- weight gradients (all three)
        if H.query("vx_jlc_wgrad_tz_ok", C, G, D, Hh, W):
            dws_old = [torch.zeros_like(w_) for w_ in ws]
            dws_new = [torch.zeros_like(w_) for w_ in ws]

            def w_old():
                H.call("vx_gconv1_bwd_weight", H.P(x), g[0].data_ptr(), H.P(dws_old[0]), None, B, C, G, D * Hh * W, st)
                H.call("vx_conv3d_bwd_weight_tiled", H.P(x), None, 0, g[1].data_ptr(), H.P(dws_old[1]), None, B, C, D, Hh, W, C, 3, 1, 1, G, 1, st)
                H.call("vx_conv3d_bwd_weight_tiled", H.P(x), None, 0, g[2].data_ptr(), H.P(dws_old[2]), None, B, C, D, Hh, W, C, 5, 1, 2, G, 1, st)

            def w_new():
                H.call("vx_jlc_wgrad_tz", H.P(x), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), H.P(dws_new[0]), H.P(dws_new[1]), H.P(dws_new[2]), B, C, G, D, Hh, W, st)

            w_old(); w_new()
            torch.cuda.synchronize()
            xg = x.double().requires_grad_(False)
            for i, k in enumerate((1, 3, 5)):
                ref = torch.nn.grad.conv3d_weight(x.double(), ws[i].shape, g[i].double(), padding=k // 2, groups=G)
                sc = float(ref.abs().max())
                print(f"  wgrad k={k}: rel-to-max err  valu {float((dws_old[i].double() - ref).abs().max()) / sc:.2e}  tz {float((dws_new[i].double() - ref).abs().max()) / sc:.2e}")
            t_old, t_new = timeit(w_old), timeit(w_new)
            print(f"  wgrad time (3 tensors): valu {t_old:.1f} us ({flops / t_old / 1e6:.1f} TFLOP/s)  tz {t_new:.1f} us ({flops / t_new / 1e6:.1f} TFLOP/s)")
            if args.breakdown:
                for mask, what in ((1, "no staging"), (2, "no MFMA phase"), (4, "no fold / atomics"), (3, "no staging, no MFMA"), (7, "launch + syncs only"), (8, "MFMA phase without its MFMAs"), (16, "MFMA phase without its g-operand reads"), (24, "MFMA phase: x windows + shifts + control only")):
                    H.call("vx_jlc_tz_set_debug", mask << 4)
                    print(f"    wgrad {what}: {timeit(w_new):.1f} us")
                H.call("vx_jlc_tz_set_debug", 0)
        if args.breakdown:
            for mask, what in ((1, "no staging"), (2, "no MFMA loops"), (3, "neither (launch + epilogue)")):
                H.call("vx_jlc_tz_set_debug", mask)
                print(f"    {what}: fwd {timeit(f_new):.1f} us  bwd {timeit(b_new):.1f} us")
            H.call("vx_jlc_tz_set_debug", 0)


if __name__ == "__main__":
    main()

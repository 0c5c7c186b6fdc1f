import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from veloxseg_amd import _hip as H
H.LIB.load()
d = torch.device("cuda")
for (B, Cin, Cout, sp) in [(4, 16, 32, (32, 32, 32)), (4, 32, 64, (16, 16, 16)), (4, 64, 128, (8, 8, 8))]:
    x = torch.randn(B, Cin, *sp, device=d); so = tuple(v // 2 for v in sp); dy = torch.randn(B, Cout, *so, device=d)
    dw = torch.zeros(Cout, Cin, 3, 3, 3, device=d)
    st = torch.cuda.current_stream().cuda_stream
    def t(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    a = t(lambda: H.call("vx_conv_wgrad_gather_mfma", H.P(x), H.P(dy), H.P(dw), None, B, Cin, *sp, Cout, 3, 2, 1, st))
    b = t(lambda: H.call("vx_conv3d_bwd_weight_tiled", H.P(x), None, 0, H.P(dy), H.P(dw), None, B, Cin, *sp, Cout, 3, 2, 1, 1, 1, st))
    print(f"Cin {Cin} Cout {Cout} {sp}: gather-GEMM {a:.1f} us, tiled VALU {b:.1f} us")

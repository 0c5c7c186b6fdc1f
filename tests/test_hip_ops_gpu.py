"""GPU parity: every HIP operator (forward + backward) against the CPU oracle on the same seeded inputs.
Calls go through the C ABI (ctypes -> libveloxseg_hip.so).  Tolerances are stated per test (fp32)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import veloxseg_oracle as O  # noqa: E402  (checker only)


def _vf():
    from veloxseg_amd import functional as VF
    return VF


def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(tuple(shape), generator=g) * scale


def close(a, b, atol, rtol, what):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    if not bool((err <= tol).all()):
        i = int((err - tol).argmax())
        raise AssertionError(f"{what}: max abs err {float(err.max()):.3e} (ref max {float(b.abs().max()):.3e}); worst idx {i}: "
                             f"got {float(a.flatten()[i]):.6e} want {float(b.flatten()[i]):.6e}; bad frac {float((err > tol).double().mean()):.3e}")


def run_pair(fn_gpu, fn_cpu, tensors, params=(), atol=2e-5, rtol=2e-4, gtol=None, seed=11, what=""):
    """tensors / params: lists of CPU tensors.  Both functions take (tensors..., params...).  Compares outputs and all grads."""
    d = dev()
    gt = [t.clone().to(d).requires_grad_(t.is_floating_point()) for t in tensors]
    gp = [t.clone().to(d).requires_grad_(True) for t in params]
    ct = [t.clone().requires_grad_(t.is_floating_point()) for t in tensors]
    cp = [t.clone().requires_grad_(True) for t in params]
    og = fn_gpu(*gt, *gp)
    oc = fn_cpu(*ct, *cp)
    og = list(og) if isinstance(og, (list, tuple)) else [og]
    oc = list(oc) if isinstance(oc, (list, tuple)) else [oc]
    assert len(og) == len(oc)
    for i, (a, b) in enumerate(zip(og, oc)):
        close(a, b, atol, rtol, f"{what} out[{i}]")
    gys = [rnd(*o.shape, seed=seed + i) for i, o in enumerate(oc)]
    torch.autograd.backward(oc, gys)
    torch.autograd.backward(og, [g.to(d) for g in gys])
    torch.cuda.synchronize()
    ga, gr = gtol or (atol * 5, rtol * 5)
    for i, (a, b) in enumerate(zip(gt, ct)):
        if b.grad is not None:
            assert a.grad is not None, f"{what}: missing grad for tensor {i}"
            close(a.grad, b.grad, ga, gr, f"{what} d tensor[{i}]")
    for i, (a, b) in enumerate(zip(gp, cp)):
        assert a.grad is not None, f"{what}: missing grad for param {i}"
        scale = max(1.0, float(b.grad.abs().max()))
        close(a.grad, b.grad, ga * scale, gr * 4, f"{what} d param[{i}]")


CONV_CASES = [
    # name, B, Cin, D,H,W, Cout, K, S, P, G, ps
    ("pw16_48", 2, 16, (6, 5, 7), 48, 1, 1, 0, 1, 1),
    ("pw_head", 2, 32, (4, 4, 4), 2, 1, 1, 0, 1, 1),
    ("jlc_k3_g4", 2, 16, (6, 6, 6), 16, 3, 1, 1, 4, 1),
    ("jlc_k5_g4", 1, 16, (7, 6, 5), 16, 5, 1, 2, 4, 1),
    ("jlc_k5_g8", 1, 64, (4, 4, 4), 64, 5, 1, 2, 8, 1),
    ("down_k7s4", 2, 2, (16, 16, 16), 16, 7, 4, 3, 1, 1),
    ("down_k3s2", 2, 16, (8, 8, 6), 32, 3, 2, 1, 1, 1),
    ("down_k3s2_c32", 3, 32, (8, 8, 8), 64, 3, 2, 1, 1, 1),          # the MFMA implicit-GEMM path (csrc/conv_mfma.hip): several column tiles, ragged row tiles
    ("down_k3s2_c64", 2, 64, (4, 4, 4), 128, 3, 2, 1, 1, 1),
    ("down_k3s2_odd_c", 1, 12, (6, 4, 10), 20, 3, 2, 1, 1, 1),
    ("stem_k7s4_c4", 1, 4, (16, 16, 12), 16, 7, 4, 3, 1, 1),
    ("down_k5s2", 1, 8, (8, 8, 8), 16, 5, 2, 2, 1, 1),
    ("embed_k4s4", 1, 1, (16, 12, 8), 16, 4, 4, 0, 1, 1),
    ("embed_k2s2", 1, 4, (8, 8, 8), 16, 2, 2, 0, 1, 1),
    ("expand_ps4", 1, 16, (4, 5, 6), 128, 3, 1, 1, 1, 4),
    ("expand_ps2", 2, 16, (4, 4, 4), 16, 3, 1, 1, 1, 2),
    ("expand_ps4_w24", 1, 16, (4, 8, 24), 128, 3, 1, 1, 1, 4),      # 24-wide rows (shipped 96^3 patches): LDS-tiled MFMA kernels with a partly idle last tile
    ("expand_ps4_w40", 2, 16, (3, 5, 40), 64, 3, 1, 1, 1, 4),       # two 32-wide chunks per row (the second one partly empty) in the split weight-gradient kernel
    ("expand_ps4_w32", 2, 16, (5, 9, 32), 128, 3, 1, 1, 1, 4),      # the headline row width; strips of 8 + 1 rows
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv3d(case):
    VF = _vf()
    name, B, Cin, sp, Cout, K, S, P, G, ps = case
    x = rnd(B, Cin, *sp)
    w = rnd(Cout, Cin // G, K, K, K, seed=1, scale=(Cin // G * K ** 3) ** -0.5)
    b = rnd(Cout, seed=2, scale=0.1)

    def g(x, w, b):
        return VF.conv3d(x, w, b, stride=S, padding=P, groups=G, pixel_shuffle=ps)

    def c(x, w, b):
        y = F.conv3d(x, w, b, stride=S, padding=P, groups=G)
        return O.pixel_shuffle3d(y, ps) if ps > 1 else y

    run_pair(g, c, [x], [w, b], what=name)


@pytest.mark.parametrize("ns", [2, 3])
@pytest.mark.parametrize("B,Cc,sp", [(2, 2, (4, 20, 32)), (1, 1, (3, 4, 24)), (2, 1, (2, 9, 72)), (1, 3, (1, 1, 4))], ids=["w32_strips", "w24", "w72_3chunks", "tiny"])
def test_expand_wgrad_split_direct(B, Cc, sp, ns):
    """C-ABI vx_expand_wgrad_mfma_split (products from ns bf16 pieces per fp32 operand, partial-sum rows folded in a fixed order) against an fp64
    reference: ns = 3 to fp32 round-off of the sums, ns = 2 to ~1e-5 relative; accumulates INTO dw / db; two runs give identical bits."""
    from veloxseg_amd import _hip as H
    d = dev()
    D_, H_, W_ = sp
    x = rnd(B, 16, *sp).to(d)
    dy = rnd(B, Cc, 4 * D_, 4 * H_, 4 * W_, seed=5).to(d)
    dyc = dy.view(B, Cc, D_, 4, H_, 4, W_, 4).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(B, 64 * Cc, D_, H_, W_)
    wref = torch.nn.grad.conv3d_weight(x.double(), (64 * Cc, 16, 3, 3, 3), dyc.double(), padding=1)
    bref = dyc.double().sum((0, 2, 3, 4))
    nws = H.query("vx_expand_wgrad_split_ws_floats", B, Cc, D_, H_, W_)
    ws = torch.full((nws,), float("nan"), device=d)                      # (every row that is folded must have been written)
    outs = []
    for _ in range(2):
        dw = torch.ones(64 * Cc, 16, 3, 3, 3, device=d)
        db = torch.full((64 * Cc,), 2.0, device=d)
        H.call("vx_expand_wgrad_mfma_split", H.P(x), H.P(dy), H.P(dw), H.P(db), H.P(ws), nws, B, Cc, D_, H_, W_, ns, H.stream_ptr())
        torch.cuda.synchronize()
        outs.append((dw.clone(), db.clone()))
    tol = 3e-6 if ns == 3 else 4e-5
    close(outs[0][0] - 1.0, wref.float(), tol * float(wref.abs().max()), tol, f"dW ns={ns}")
    close(outs[0][1] - 2.0, bref.float(), 3e-6 * float(bref.abs().max()) + 1e-5, 1e-5, "db")
    assert torch.equal(outs[0][0], outs[1][0]), "the folded weight gradient is not reproducible"


@pytest.mark.parametrize("ns", [3, 22, 2])
@pytest.mark.parametrize("B,Cc,sp,scale", [(2, 2, (4, 8, 32), 1.0), (1, 1, (4, 4, 24), 1.0), (2, 1, (8, 4, 16), 3.0e4), (1, 3, (4, 4, 4), 2.0e-5)],
                         ids=["w32", "w24_ragged_tile", "large_values", "tiny_values"])
def test_expand_fwd_and_input_gradient_split_direct(B, Cc, sp, ns, scale):
    """C-ABI vx_expand_fwd_mfma_split / vx_expand_bwd_data_mfma_split against an fp64 reference.  ns = 3: three bf16 pieces; ns = 22: two fp16 pieces of the operand
    scaled by a power of two per staged tile (and per weight tensor), exact rescale of the fp32 accumulators -- both to fp32 round-off of the sums; ns = 2 to ~1e-5.
    The large / tiny cases put the activations outside fp16's range: the scaling must bring them back."""
    from veloxseg_amd import _hip as H
    d = dev()
    D_, H_, W_ = sp
    x = (rnd(B, 16, *sp) * scale).to(d)
    w = rnd(64 * Cc, 16, 3, 3, 3, seed=1, scale=(16 * 27) ** -0.5).to(d)
    b = rnd(64 * Cc, seed=2, scale=0.1 * scale).to(d)
    dy = (rnd(B, Cc, 4 * D_, 4 * H_, 4 * W_, seed=5) * scale).to(d)
    nws = max(64 * Cc * 16 * 27, H.query("vx_expand_split_ws_floats", Cc, ns))
    ws = torch.empty(nws, device=d)
    y = torch.full((B, Cc, 4 * D_, 4 * H_, 4 * W_), float("nan"), device=d)
    rc = H.query("vx_expand_fwd_mfma_split", H.P(x), H.P(w), H.P(b), H.P(ws), H.P(y), B, Cc, D_, H_, W_, ns, H.stream_ptr())
    assert rc == 0
    yref = O.pixel_shuffle3d(F.conv3d(x.double(), w.double(), b.double(), padding=1), 4)
    tol = {2: 4e-5, 3: 4e-6, 22: 1.5e-6}[ns]                 # (measured: 5e-6 / 1.4e-6 / 4e-7; torch's own fp32 convolution: 3e-7)
    assert float((y.double() - yref).abs().max()) <= tol * float(yref.abs().max()), f"forward ns={ns}"
    dx = torch.full((B, 16, *sp), float("nan"), device=d)
    rc = H.query("vx_expand_bwd_data_mfma_split", H.P(dy), H.P(w), H.P(ws), H.P(dx), B, Cc, D_, H_, W_, 0, ns, H.stream_ptr())
    assert rc == 0
    dyc = dy.view(B, Cc, D_, 4, H_, 4, W_, 4).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(B, 64 * Cc, D_, H_, W_)
    dxref = F.conv_transpose3d(dyc.double(), w.double(), None, padding=1)
    assert float((dx.double() - dxref).abs().max()) <= tol * float(dxref.abs().max()), f"input gradient ns={ns}"
    keep = dx.clone()
    rc = H.query("vx_expand_bwd_data_mfma_split", H.P(dy), H.P(w), H.P(ws), H.P(dx), B, Cc, D_, H_, W_, 1, ns, H.stream_ptr())      # accumulate = 1
    assert rc == 0 and float((dx.double() - 2 * keep.double()).abs().max()) <= 1e-6 * float(dxref.abs().max())


def test_conv3d_concat_nobias():
    VF = _vf()
    x1, x2 = rnd(2, 16, 4, 4, 4), rnd(2, 32, 4, 4, 4, seed=3)
    w = rnd(24, 48, 1, 1, 1, seed=4, scale=0.15)
    run_pair(lambda a, b, w: VF.conv3d(a, w, None, x2=b), lambda a, b, w: F.conv3d(torch.cat([a, b], 1), w), [x1, x2], [w], what="concat")


@pytest.mark.parametrize("C1,C2,Co,concat", [(16, 16, 16, True), (64, 0, 16, False), (32, 0, 32, False)], ids=["concat_16_16_to_16", "64_to_16", "32_to_32"])
def test_conv3d_1x1_large_volume_kernel(C1, C2, Co, concat):
    """1x1 convolutions of the 32^3 level (V = 32768 per sample: the one-wave kernel with 16-byte accesses and the weight tile in LDS, pointwise.hip vx_pw_fwd_v4_k)
    against torch, forward and every gradient, with and without the in-kernel channel concat; and against the one-voxel-per-thread kernel of the same library
    (vx_pw_conv_set_v4(0)): same products, fp32 summation noise."""
    VF = _vf()
    from veloxseg_amd import _hip as H
    x1 = rnd(2, C1, 32, 32, 32)
    w = rnd(Co, C1 + C2, 1, 1, 1, seed=4, scale=(C1 + C2) ** -0.5)
    b = rnd(Co, seed=5, scale=0.1)
    if concat:
        x2 = rnd(2, C2, 32, 32, 32, seed=3)
        run_pair(lambda a, c, w, b: VF.conv3d(a, w, b, x2=c), lambda a, c, w, b: F.conv3d(torch.cat([a, c], 1), w, b), [x1, x2], [w, b], what="1x1 v4 concat")
    else:
        run_pair(lambda a, w, b: VF.conv3d(a, w, b), lambda a, w, b: F.conv3d(a, w, b), [x1], [w, b], what="1x1 v4")
    d = dev()
    outs = []
    try:
        for on in (1, 0):
            H.call("vx_pw_conv_set_v4", on)
            with torch.no_grad():
                outs.append(VF.conv3d(x1.to(d), w.to(d), b.to(d), x2=(x2.to(d) if concat else None)))
    finally:
        H.call("vx_pw_conv_set_v4", 1)
    close(outs[0], outs[1], 2e-6 * float(outs[1].abs().max()), 1e-5, "v4 kernel vs one-voxel-per-thread kernel")


@pytest.mark.parametrize("B,Cin,sp,scale", [(2, 2, (16, 16, 64), 1.0), (1, 1, (8, 32, 128), 1.0), (1, 4, (8, 16, 64), 1.0), (1, 2, (12, 16, 64), 2.0e4), (1, 2, (8, 16, 64), 1.0e-5)],
                         ids=["two_modalities", "one_channel_wide", "four_channels", "large_values", "tiny_values"])
def test_stem_conv_on_the_f16_pipe(B, Cin, sp, scale):
    """The stem convolution (k = 7, s = 4, p = 3, 16 channels: conv_mfma.hip vx_stem_fwd_k, input rows staged once in LDS as two scaled fp16 pieces, Toeplitz GEMM along
    W) against torch's fp64 convolution and against the fp32 gather kernel it replaces (vx_conv_mfma_set_stem_f16(0)): error at the level of the fp32 kernel's."""
    VF = _vf()
    from veloxseg_amd import _hip as H
    d = dev()
    x = (rnd(B, Cin, *sp) * scale).to(d)
    w = rnd(16, Cin, 7, 7, 7, seed=1, scale=(Cin * 343) ** -0.5).to(d)
    b = (rnd(16, seed=2, scale=0.1) * scale).to(d)
    ref = F.conv3d(x.double(), w.double(), b.double(), stride=4, padding=3)
    outs = []
    try:
        for on in (2, 0):          # (2: also with four input channels, which the default rule leaves to the fp32 kernel)
            H.call("vx_conv_mfma_set_stem_f16", on)
            with torch.no_grad():
                outs.append(VF.conv3d(x, w, b, stride=4, padding=3))
    finally:
        H.call("vx_conv_mfma_set_stem_f16", 1)          # (the library's default since round 5: see csrc/conv_mfma.hip)
    sc = float(ref.abs().max())
    e_new, e_old = float((outs[0].double() - ref).abs().max()) / sc, float((outs[1].double() - ref).abs().max()) / sc
    assert e_new <= max(3.0 * e_old, 2e-6), (e_new, e_old)


@pytest.mark.parametrize("B,Cin,sp,scale", [(2, 2, (16, 32, 128), 1.0), (1, 1, (8, 16, 128), 1.0), (1, 4, (12, 16, 128), 1.0), (1, 2, (8, 32, 128), 3.0e4), (1, 2, (8, 16, 128), 2.0e-5),
                                            (2, 2, (8, 32, 96), 1.0), (1, 4, (12, 16, 96), 1.0)],
                         ids=["two_modalities", "one_channel", "four_channels", "large_values", "tiny_values", "rows_of_96", "rows_of_96_four_channels"])
def test_stem_weight_gradient_on_the_f16_pipe(B, Cin, sp, scale):
    """The stem convolution's weight / bias gradient (conv_wgrad.hip vx_stem_wgrad_f16_k: de-interleaved input rows in LDS, two scaled fp16 pieces per operand, one
    16x16x32 MFMA per output row and 16 taps) against torch's fp64 gradient and against the fp32-MFMA kernel it replaces (vx_down_wgrad_set_f16(0)), through the C ABI;
    accumulation into a non-zero dw / db as the engine's flat gradient buffer requires."""
    from veloxseg_amd import _hip as H
    d = dev()
    x = (rnd(B, Cin, *sp) * scale).to(d)
    Do, Ho, Wo = sp[0] // 4, sp[1] // 4, sp[2] // 4
    dy = (rnd(B, 16, Do, Ho, Wo, seed=3) * (0.5 * scale)).to(d)
    w = rnd(16, Cin, 7, 7, 7, seed=1).to(d).double().requires_grad_(True)
    bb = torch.zeros(16, device=d, dtype=torch.float64, requires_grad=True)
    F.conv3d(x.double(), w, bb, stride=4, padding=3).backward(dy.double())
    ref_w, ref_b = w.grad, bb.grad
    nws = H.query("vx_down_wgrad_ws_floats", B, Cin, *sp, 16)
    assert nws > 0
    outs = []
    try:
        for on in (1, 0):
            H.call("vx_down_wgrad_set_f16", on)
            dw = torch.full((16, Cin, 7, 7, 7), 0.25, device=d)
            db = torch.full((16,), -0.5, device=d)
            ws = torch.empty(nws, device=d)
            H.call("vx_down_wgrad_mfma", H.P(x), H.P(dy), H.P(dw), H.P(db), H.P(ws), nws, B, Cin, *sp, 16, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            outs.append((dw.double() - 0.25, db.double() + 0.5))
    finally:
        H.call("vx_down_wgrad_set_f16", 1)
    sc = float(ref_w.abs().max())
    e_new, e_old = float((outs[0][0] - ref_w).abs().max()) / sc, float((outs[1][0] - ref_w).abs().max()) / sc
    assert e_new <= max(3.0 * e_old, 3e-6), (e_new, e_old)
    scb = float(ref_b.abs().max())
    assert float((outs[0][1] - ref_b).abs().max()) <= 2e-5 * scb + 2.5e-7          # (db is accumulated into a buffer that holds -0.5: fp32 rounding there)


@pytest.mark.parametrize("B,Cin,sp", [(2, 2, (64, 64, 128)), (1, 1, (32, 48, 128))], ids=["two_channels", "one_channel"])
def test_stem_forward_hands_max_abs_x_to_its_weight_gradient(B, Cin, sp):
    """vx_conv_mfma_fwd_mx leaves the bits of max |x| (a by-product of the stem kernel's staging); vx_down_wgrad_mfma_mx with it == vx_down_wgrad_mfma that reads x to
    find it: bit-identical dw / db, and the forward output is the one of the plain entry"""
    from veloxseg_amd import _hip as H
    d = dev()
    x = (rnd(B, Cin, *sp) * 3.0).to(d)
    x[0, 0, 5, 7, 9] = -41.5                                   # (the maximum is a negative element well inside a tile)
    w, bias = (rnd(16, Cin, 7, 7, 7, seed=1) * 0.1).to(d), rnd(16, seed=2).to(d)
    Do, Ho, Wo = sp[0] // 4, sp[1] // 4, sp[2] // 4
    dy = rnd(B, 16, Do, Ho, Wo, seed=3).to(d)
    st = H.stream_ptr()
    assert H.query("vx_conv_mfma_fwd_writes_absmax", Cin, 16, *sp, 7, 4, 3) == 1
    nf = H.query("vx_conv_mfma_ws_floats", Cin, 16, 7, 0)
    y0, y1 = torch.empty(B, 16, Do, Ho, Wo, device=d), torch.empty(B, 16, Do, Ho, Wo, device=d)
    wsf = torch.empty(nf, device=d)
    mx = torch.full((1,), 12345, device=d, dtype=torch.int32)
    H.call("vx_conv_mfma_fwd", H.P(x), H.P(w), H.P(bias), H.P(y0), H.P(wsf), B, Cin, *sp, 16, 7, 4, 3, st)
    H.call("vx_conv_mfma_fwd_mx", H.P(x), H.P(w), H.P(bias), H.P(y1), H.P(wsf), mx.data_ptr(), B, Cin, *sp, 16, 7, 4, 3, st)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert mx.view(torch.float32).item() == 41.5
    nws = H.query("vx_down_wgrad_ws_floats", B, Cin, *sp, 16)
    outs = []
    for hand in (False, False, True):
        dw, db, ws = torch.full((16, Cin, 7, 7, 7), 0.25, device=d), torch.full((16,), -0.5, device=d), torch.empty(nws, device=d)
        if hand:
            H.call("vx_down_wgrad_mfma_mx", H.P(x), H.P(dy), H.P(dw), H.P(db), H.P(ws), nws, mx.data_ptr(), B, Cin, *sp, 16, st)
        else:
            H.call("vx_down_wgrad_mfma", H.P(x), H.P(dy), H.P(dw), H.P(db), H.P(ws), nws, B, Cin, *sp, 16, st)
        torch.cuda.synchronize()
        outs.append((dw, db))
    # (the same scales -> the same operand pieces; the sums themselves are folded with float atomics where the kernel says so: bit-identity only where two plain runs agree)
    if torch.equal(outs[0][0], outs[1][0]):
        assert torch.equal(outs[0][0], outs[2][0])
    if torch.equal(outs[0][1], outs[1][1]):
        assert torch.equal(outs[0][1], outs[2][1])
    sc = float(outs[0][0].abs().max())
    rr = float((outs[0][0] - outs[1][0]).abs().max()) / sc               # run-to-run spread of the plain entry
    assert float((outs[0][0] - outs[2][0]).abs().max()) / sc <= max(4.0 * rr, 1e-6), rr
    close(outs[2][1], outs[0][1], 1e-5 * float(outs[0][1].abs().max()), 1e-5, "db")


@pytest.mark.parametrize("B,Cin,Cout,sp", [(4, 16, 32, (32, 32, 32)), (2, 32, 64, (16, 16, 16)), (4, 64, 128, (8, 8, 8)), (1, 16, 32, (24, 24, 24)), (2, 32, 64, (12, 12, 12)), (2, 64, 128, (6, 6, 6))],
                         ids=["down2_128", "down3_128", "down4_128", "down2_96", "down3_96", "down4_96"])
def test_downconv_weight_gradient_gather_gemm(B, Cin, Cout, sp):
    """The weight gradient of the level 2 - 4 DownConvs (Conv3d k3 s2 p1, conv_blocks.py:4-21) as a gather-GEMM on the fp32 matrix pipe (conv_wgrad.hip
    vx_wgrad_gather_mfma_k) against torch's fp64 gradient and the tiled VALU kernel it replaces, through the C ABI; accumulation into a non-zero dw."""
    from veloxseg_amd import _hip as H
    d = dev()
    x = rnd(B, Cin, *sp).to(d)
    so = tuple((v + 2 - 3) // 2 + 1 for v in sp)
    dy = rnd(B, Cout, *so, seed=4).to(d)
    w = rnd(Cout, Cin, 3, 3, 3, seed=1).to(d).double().requires_grad_(True)
    bb = torch.zeros(Cout, device=d, dtype=torch.float64, requires_grad=True)
    F.conv3d(x.double(), w, bb, stride=2, padding=1).backward(dy.double())
    ref, ref_b = w.grad, bb.grad
    st = torch.cuda.current_stream().cuda_stream
    assert H.query("vx_conv_wgrad_gather_ok", B, Cin, *sp, Cout, 3, 2, 1) == 1
    dw_new = torch.full((Cout, Cin, 3, 3, 3), 0.5, device=d)
    db_new = torch.full((Cout,), 2.0, device=d)
    H.call("vx_conv_wgrad_gather_mfma", H.P(x), H.P(dy), H.P(dw_new), H.P(db_new), B, Cin, *sp, Cout, 3, 2, 1, st)
    dw_old = torch.full((Cout, Cin, 3, 3, 3), 0.5, device=d)
    H.call("vx_conv3d_bwd_weight_tiled", H.P(x), None, 0, H.P(dy), H.P(dw_old), None, B, Cin, *sp, Cout, 3, 2, 1, 1, 1, st)
    torch.cuda.synchronize()
    sc = float(ref.abs().max())
    e_new, e_old = float((dw_new.double() - 0.5 - ref).abs().max()) / sc, float((dw_old.double() - 0.5 - ref).abs().max()) / sc
    assert e_new <= max(3.0 * e_old, 3e-6), (e_new, e_old)
    assert float((db_new.double() - 2.0 - ref_b).abs().max()) <= 2e-5 * float(ref_b.abs().max()) + 1e-6


@pytest.mark.parametrize("B,C,G,sp", [(4, 64, 8, (8, 8, 8)), (4, 128, 8, (4, 4, 4)), (2, 64, 8, (6, 6, 6)), (2, 128, 8, (3, 3, 3)), (1, 32, 4, (6, 6, 6)), (2, 16, 4, (4, 6, 8))],
                         ids=["level3_128", "level4_128", "level3_96", "level4_96", "group_of_8_small", "group_of_4_aniso"])
def test_jlc_weight_gradients_small_volume_gather_gemm(B, C, G, sp):
    """The three weight gradients of a JLC block (grouped k = 5 / 3 / 1 convolutions, conv_blocks.py:51-58) at small volumes in one gather-GEMM launch (conv_wgrad.hip
    vx_jlc_wgrad_gather_k: the 8^3 / 4^3 levels at 128^3, and the 6^3 / 3^3 levels of the shipped 96^3 configurations that the Toeplitz kernel does not cover) against
    torch's fp64 gradients, through the C ABI; accumulation into non-zero buffers."""
    from veloxseg_amd import _hip as H
    d = dev()
    CG = C // G
    x = rnd(B, C, *sp).to(d)
    gs = [rnd(B, C, *sp, seed=10 + k).to(d) for k in (1, 3, 5)]
    refs = []
    for g_, k in zip(gs, (1, 3, 5)):
        w = torch.zeros(C, CG, k, k, k, device=d, dtype=torch.float64, requires_grad=True)
        F.conv3d(x.double(), w, None, padding=k // 2, groups=G).backward(g_.double())
        refs.append(w.grad)
    assert H.query("vx_jlc_wgrad_gather_ok", C, G, *sp) == 1
    dws = [torch.full((C, CG, k, k, k), 0.125, device=d) for k in (1, 3, 5)]
    H.call("vx_jlc_wgrad_gather", H.P(x), H.P(gs[0]), H.P(gs[1]), H.P(gs[2]), H.P(dws[0]), H.P(dws[1]), H.P(dws[2]), B, C, G, *sp, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for dw, ref, k in zip(dws, refs, (1, 3, 5)):
        sc = float(ref.abs().max())
        assert float((dw.double() - 0.125 - ref).abs().max()) <= 3e-6 * sc + 1e-6, (k, float((dw.double() - 0.125 - ref).abs().max()), sc)


def test_fan_out_gradients_meet_in_one_sum():
    """functional.fan_out (one alias of a tensor per consumer, the consumers' gradients summed in one vx_add_many launch for the whole list) against plain autograd
    accumulation: same forward values, same gradients, for two tensors with three consumers each and one with two."""
    VF = _vf()
    d = dev()
    xs = [rnd(2, 8, 4, 4, 4, seed=i).to(d).requires_grad_(True) for i in range(2)]
    ys = [x.detach().clone().requires_grad_(True) for x in xs]
    ws = [rnd(2, 8, 4, 4, 4, seed=10 + i).to(d) for i in range(3)]
    al = VF.fan_out([x * 1.0 for x in xs], 3)
    assert all(torch.equal(a[j], xs[i].detach()) for i, a in enumerate(al) for j in range(3))
    loss = sum((al[i][j] * ws[j]).sum() * (i + 1) for i in range(2) for j in range(3))
    ref = sum((y * ws[j]).sum() * (i + 1) for i, y in enumerate(ys) for j in range(3))
    loss.backward()
    ref.backward()
    for x, y in zip(xs, ys):
        close(x.grad, y.grad, 1e-5, 1e-5, "fan_out gradient")
    (a2,) = VF.fan_out([xs[0] * 2.0], 2)
    g = torch.autograd.grad((a2[0] * ws[0]).sum() + (a2[1] * ws[1]).sum(), xs[0])[0]
    close(g, 2.0 * (ws[0] + ws[1]), 1e-5, 1e-5, "fan_out of two")


@pytest.mark.parametrize("B,Ci,Co,sp", [(2, 32, 16, (3, 4, 5)), (4, 128, 64, (4, 4, 4)), (3, 64, 32, (8, 8, 8)), (2, 32, 16, (16, 16, 16)), (1, 24, 8, (3, 3, 3)),
                                        (2, 48, 24, (3, 5, 8)), (1, 20, 6, (2, 3, 12))],
                         ids=["ragged", "L4_to_L3", "L3_to_L2", "L2_to_L1", "odd_channels", "rows_of_8_ragged_tiles", "rows_of_12_partial_row_tiles"])
def test_conv_transpose(B, Ci, Co, sp):
    """ConvTranspose3d(k2, s2): forward, input gradient, and the weight gradient as one MFMA GEMM (pointwise.hip vx_upconv_k2s2_wgrad; channel counts that
    are not multiples of 16 keep the generic kernel).  Rows of 4 k voxels run the 16 x 64 tiles with 16-byte accesses (vx_upconv_mfma4_k: partial row / voxel
    tiles, channel counts that are not multiples of 16), other widths the 16 x 16 tiles."""
    VF = _vf()
    x = rnd(B, Ci, *sp)
    w = rnd(Ci, Co, 2, 2, 2, seed=5, scale=0.2)
    b = rnd(Co, seed=6, scale=0.1)
    run_pair(lambda x, w, b: VF.conv_transpose_k2s2(x, w, b), lambda x, w, b: F.conv_transpose3d(x, w, b, stride=2), [x], [w, b], what="convT")


@pytest.mark.parametrize("shape", [(2, 8, 5, 4, 3), (2, 4, 16, 16, 16), (1, 3, 20, 20, 12)], ids=["V60_row", "V4096_row", "V4800_split"])
@pytest.mark.parametrize("n,act,res", [(1, False, False), (1, False, True), (2, False, False), (3, True, True)])
def test_instnorm_sum(n, act, res, shape):
    """short rows (V <= 4096) run the one-launch row kernels, longer ones the split statistics + apply kernels; same oracle"""
    VF = _vf()
    ys = [rnd(*shape, seed=i) * (1 + i) + i for i in range(n)]
    r = [rnd(*shape, seed=9)] if res else []

    def g(*t):
        return VF.instnorm_sum(list(t[:n]), act=act, res=t[n] if res else None)

    def c(*t):
        out = t[n] if res else 0
        for y in t[:n]:
            z = O.instnorm(y)
            out = out + (O.gelu(z) if act else z)
        return out

    run_pair(g, c, ys + r, what=f"in{n}")
    if shape[2] * shape[3] * shape[4] <= 4096:          # and the two kernel families agree with each other
        import veloxseg_amd.functional as F_
        d = dev()
        outs = {}
        try:
            for flag in (True, False):
                F_.USE_IN_ROW = flag
                t = [y.clone().to(d).requires_grad_(True) for y in ys + r]
                o = g(*t)
                o.backward(rnd(*shape, seed=77).to(d))
                outs[flag] = [o.detach()] + [x.grad for x in t]
        finally:
            F_.USE_IN_ROW = True
        for a, b in zip(outs[True], outs[False]):
            close(a, b, 1e-5 * max(1.0, float(b.abs().max())), 1e-5, "row vs split")


def test_layernorm_and_s2d():
    VF = _vf()
    x = rnd(2, 16, 4, 6, 8)
    run_pair(lambda x, w, b: VF.layernorm_cf(x, w, b), lambda x, w, b: O.layernorm_cf(x, w, b), [x], [1 + 0.2 * rnd(16, seed=1), 0.2 * rnd(16, seed=2)], what="ln")
    run_pair(lambda x: VF.space_to_depth2(x), lambda x: O.space_to_depth2(x), [x], atol=0, rtol=0, what="s2d")
    # few voxels, many channels (PatchMerging LN over 8C at the coarse levels): the lane-parallel kernels
    for shape in [(2, 128, 4, 4, 4), (1, 64, 3, 5, 2), (4, 512, 4, 4, 4), (2, 100, 2, 3, 3), (3, 256, 5, 7, 9), (2, 1024, 2, 2, 2), (4, 128, 16, 16, 16)]:     # (4, 512, 4^3), (3, 256, ..), (2, 1024, ..): 64 lanes per voxel
        C = shape[1]
        run_pair(lambda x, w, b: VF.layernorm_cf(x, w, b), lambda x, w, b: O.layernorm_cf(x, w, b), [rnd(*shape, seed=C) * 2 + 0.5],
                 [1 + 0.2 * rnd(C, seed=1), 0.2 * rnd(C, seed=2)], gtol=(2e-5, 2e-3), what=f"ln{shape}")


def test_elementwise():
    VF = _vf()
    a, z = rnd(2, 6, 4, 4, 4), rnd(2, 6, 4, 4, 4, seed=3)
    run_pair(lambda a: VF.gelu_dropout(a), lambda a: O.gelu(a), [a], what="gelu")
    run_pair(lambda a, z: VF.residual_dropout(a, z, 2.0), lambda a, z: 2 * a + z, [a, z], what="axpy2")
    run_pair(lambda a, z: VF.residual_dropout(a, z, 1.0), lambda a, z: a + z, [a, z], what="axpy1")
    run_pair(lambda a, z: VF.add(a, z), lambda a, z: a + z, [a, z], what="add")


def test_dropout_statistics_and_bwd_mask():
    VF = _vf()
    d = dev()
    VF.manual_seed(777, d)
    z = torch.ones(1 << 20, device=d).view(1, 1, 64, 128, 128).requires_grad_(True)
    out = VF.residual_dropout(None, z, 0.0, 0.1, 5)
    keep = float((out != 0).float().mean())
    assert abs(keep - 0.9) < 3e-3, keep
    assert abs(float(out.max()) - 1 / 0.9) < 1e-6
    out.sum().backward()
    assert torch.equal((z.grad != 0), (out != 0)), "backward must regenerate the forward mask"
    out2 = VF.residual_dropout(None, z.detach(), 0.0, 0.1, 6)
    assert float(((out2 != 0) != (out != 0)).float().mean()) > 0.1, "different sites must give different masks"
    VF.advance_rng(d)
    out3 = VF.residual_dropout(None, z.detach(), 0.0, 0.1, 5)
    assert float(((out3 != 0) != (out != 0)).float().mean()) > 0.1, "a new step must give a new mask"


PWA_CASES = [
    # grid, big, heads, min_dim_head, C, M
    ("l2_like", [12, 12, 12], [6, 6, 6], 2, 8, 32, 2),
    ("l1_like", [8, 8, 8], [2, 2, 2], 1, 4, 16, 2),
    ("single_modality", [8, 8, 8], [4, 4, 4], 2, 8, 32, 1),
    ("aniso", [8, 8, 4], [4, 4, 2], 1, 4, 16, 2),
    ("l4_like", [3, 3, 3], [3, 3, 3], 4, 16, 128, 2),
]


@pytest.mark.parametrize("case", PWA_CASES, ids=[c[0] for c in PWA_CASES])
def test_pwa_core(case):
    VF = _vf()
    from veloxseg_amd import _hip as H
    name, grid, big, heads, mdh, C, M = case
    pl = O.plan_pwa(grid, big, [1, 1, 1], 2, heads, mdh, C)
    plan = H.make_plan(grid, pl["n"], heads, pl["small"], pl["nwin"])
    n = pl["n"]
    table = rnd((2 * n[0] - 1) * (2 * n[1] - 1) * (2 * n[2] - 1), heads, seed=4, scale=0.5)
    idx = O.relative_position_index(n)
    qkv = []
    for m in range(M):
        qkv += [rnd(2, pl["ch_qk"], *grid, seed=10 + m), rnd(2, pl["ch_qk"], *grid, seed=20 + m), rnd(2, pl["ch_v"], *grid, seed=30 + m)]

    def g(*t):
        return VF.pwa_core(t[-1], plan, pl["c_qk"], pl["c_v"], list(t[:-1]))

    def c(*t):
        qs = [O.gather_windows(t[3 * m], pl, pl["c_qk"]) for m in range(M)]
        ks = [O.gather_windows(t[3 * m + 1], pl, pl["c_qk"]) for m in range(M)]
        vs = [O.gather_windows(t[3 * m + 2], pl, pl["c_v"]) for m in range(M)]
        l = qs[0].shape[3]
        a = O.window_attention(torch.cat(qs, 3), torch.cat(ks, 3), torch.cat(vs, 3), O.relative_bias(t[-1], idx, l), M)
        return [O.scatter_windows(a[:, :, :, m * l:(m + 1) * l], pl, pl["c_v"]) for m in range(M)]

    run_pair(g, c, qkv, [table], atol=3e-5, rtol=3e-4, what=name)


@pytest.mark.parametrize("grid,big,heads,mdh,C", [([12, 12, 12], [6, 6, 6], 2, 8, 32), ([6, 6, 6], [3, 3, 3], 1, 4, 16), ([16, 16, 16], [8, 8, 8], 2, 8, 32)],
                         ids=["ML432", "ML54_unaligned", "ML1024"])
def test_pwa_attention_dropout_and_key_splits(grid, big, heads, mdh, C):
    """Attention dropout (p = 0.3) and the key/query split of the attention kernels.
    (a) S = 1, 2, 4 give the same outputs and gradients (the mask is a function of (seed, step, site, element) only);
    (b) forward and backward use the same mask: <O(V), G> == <V, dV(G)> (O is linear in V) and a central difference along a random
        direction of q matches <dq, direction>;
    (c) the mask keeps ~70 % of the pairs: with q = k = 0, v = 1 every output is the mean of its row's mask (expectation 1)."""
    VF = _vf()
    from veloxseg_amd import _hip as H
    d = dev()
    M = 2
    pl = O.plan_pwa(grid, big, [1, 1, 1], 2, heads, mdh, C)
    plan = H.make_plan(grid, pl["n"], heads, pl["small"], pl["nwin"])
    n = pl["n"]
    table = (rnd((2 * n[0] - 1) * (2 * n[1] - 1) * (2 * n[2] - 1), heads, seed=4, scale=0.5)).to(d).requires_grad_(True)
    base = []
    for m in range(M):
        base += [rnd(2, pl["ch_qk"], *grid, seed=10 + m), rnd(2, pl["ch_qk"], *grid, seed=20 + m), rnd(2, pl["ch_v"], *grid, seed=30 + m)]
    gouts = None

    def run(S, tensors=None, backward=True):
        nonlocal gouts
        H.call("vx_pwa_attn_set_split", S)
        VF.manual_seed(77, d)
        t = [b.clone().to(d).requires_grad_(True) for b in (tensors or base)]
        table.grad = None
        outs = VF.pwa_core(table, plan, pl["c_qk"], pl["c_v"], t, p_attn=0.3, site=9)
        if gouts is None:
            gouts = [rnd(*o.shape, seed=50 + i).to(d) for i, o in enumerate(outs)]
        if backward:
            torch.autograd.backward(outs, gouts)
        torch.cuda.synchronize()
        return [o.detach() for o in outs], [x.grad for x in t] + [table.grad.clone()] if backward else None

    try:
        o1, g1 = run(1)
        for S in (2, 4):
            oS, gS = run(S)
            for i, (a, b) in enumerate(zip(oS + gS, o1 + g1)):
                close(a, b, 2e-5 * max(1.0, float(b.abs().max())), 1e-4, f"split {S} tensor {i}")
        H.call("vx_pwa_attn_set_split", 0)
        o0, g0 = run(0)
        # (b) adjoint in v (modality 0) and directional derivative in q (modality 1)
        zero_v = [b.clone() for b in base]
        zero_v[5] = torch.zeros_like(base[5])                    # O = A(v0) + B(v1), both positively homogeneous (max-pool + linear maps)
        oz, _ = run(0, zero_v, backward=False)                   # => <A(v0), G> = <v0, dv0> (Euler), A(v0) = O(v0, 0)
        lhs0 = sum(float((o.double() * g.double()).sum()) for o, g in zip(oz, gouts))
        rhs0 = float((base[2].to(d).double() * g0[2].double()).sum())
        assert abs(lhs0 - rhs0) <= 2e-4 * max(1.0, abs(rhs0)), ("v adjoint", lhs0, rhs0)
        dirq = rnd(*base[3].shape, seed=91)
        eps = 3e-3          # small: every arg-max switch of the max-pooled window scales that the step crosses is a kink of the forward map
        vals = []
        for sgn in (+1, -1):
            tt = [b.clone() for b in base]
            tt[3] = base[3] + sgn * eps * dirq
            oo, _ = run(0, tt, backward=False)
            vals.append(sum(float((o.double() * g.double()).sum()) for o, g in zip(oo, gouts)))
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = float((g0[3].double() * dirq.to(d).double()).sum())
        assert abs(fd - an) <= 1e-1 * max(1.0, abs(an)), ("q directional derivative", fd, an)     # (a mask mismatch between forward and dQ pass shows as an O(1) error)
        # (c) mask statistics
        flat = [torch.zeros_like(b) if i % 3 != 2 else torch.ones_like(b) for i, b in enumerate(base)]
        table0 = table.detach() * 0
        VF.manual_seed(78, d)
        tq = VF.pwa_core(table0, plan, pl["c_qk"], pl["c_v"], [f.to(d) for f in flat], p_attn=0.3, site=9)
        mean = float(torch.stack([t.mean() for t in tq]).mean())
        assert abs(mean - 1.0) < 2e-2, mean
        assert float(tq[0].std()) > 1e-3, "dropout must perturb the rows"
    finally:
        H.call("vx_pwa_attn_set_split", 0)


def test_upsample_and_gram():
    VF = _vf()
    x = rnd(2, 3, 4, 6, 3)
    run_pair(lambda x: VF.upsample_trilinear(x, (16, 24, 12)), lambda x: O.upsample_trilinear(x, [16, 24, 12]), [x], what="up")
    x = rnd(1, 2, 2, 2, 2)
    run_pair(lambda x: VF.upsample_trilinear(x, (32, 32, 32)), lambda x: O.upsample_trilinear(x, [32, 32, 32]), [x], what="up16x")
    x = rnd(2, 16, 6, 6, 6)
    run_pair(lambda x: VF.gram(x), lambda x: O.gram(x), [x], atol=1e-6, rtol=1e-4, what="gram")
    x = rnd(2, 8, 5, 5, 5)
    run_pair(lambda x: VF.gram(x), lambda x: O.gram(x), [x], atol=1e-6, rtol=1e-4, what="gram8")


@pytest.mark.parametrize("ncls,M,labdtype", [(2, 2, torch.int64), (4, 1, torch.uint8), (3, 2, torch.int32)])
def test_loss(ncls, M, labdtype):
    VF = _vf()
    d = dev()
    B, S = 2, 8
    g = torch.Generator().manual_seed(5)
    outs = [rnd(B, ncls, S, S, S, seed=i) for i in range(4)] + [rnd(B, M, S, S, S, seed=7)] + [rnd(B, 16, 16, seed=8 + i, scale=0.1) for i in range(1 + M)]
    lab = torch.randint(0, ncls, (B, 1, S, S, S), generator=g)
    sr = rnd(B, M, S, S, S, seed=99)
    cfg = {"deep_Loss_weight": [1, 1, 1, 1], "RC_Loss_weight": 0.5, "Feature_Loss_weight": 2.0}

    def gfn(*t):
        return VF.veloxseg_loss(list(t), lab.to(d).to(labdtype), sr.to(d), [0.25] * 4, 0.5, 2.0, M)

    def cfn(*t):
        return O.loss(list(t), lab, sr, M, cfg)

    run_pair(gfn, cfn, outs, atol=1e-5, rtol=1e-5, gtol=(1e-8, 1e-3), what=f"loss{ncls}")


def test_adamw_matches_torch():
    from veloxseg_amd import _hip as H
    d = dev()
    n = 10007
    p0, g0 = rnd(n), rnd(n, seed=1)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=2.5e-4, weight_decay=0.01)
    p = p0.clone().to(d)
    m, v = torch.zeros(n, device=d), torch.zeros(n, device=d)
    for step in range(1, 4):
        gi = g0 * step
        pt.grad = gi.clone()
        opt.step()
        H.call("vx_adamw_step", H.P(p), H.P(gi.to(d)), H.P(m), H.P(v), n, 2.5e-4, 0.9, 0.999, 1e-8, 0.01, step, 1.0, H.stream_ptr())
    close(p, pt.data, 1e-7, 1e-6, "adamw")


def test_errors_are_python_exceptions():
    VF = _vf()
    d = dev()
    with pytest.raises(RuntimeError, match="no CPU"):
        VF.conv3d(torch.zeros(1, 4, 4, 4, 4), torch.zeros(4, 4, 1, 1, 1))
    with pytest.raises(RuntimeError, match="InstanceNorm"):
        VF.instnorm_sum([torch.zeros(1, 4, 1, 1, 1, device=d)])
    from veloxseg_amd import _hip as H
    with pytest.raises(RuntimeError, match="vx_conv3d_fwd"):
        H.call("vx_conv3d_fwd", None, None, 0, None, None, None, 1, 4, 4, 4, 4, 4, 1, 1, 0, 1, 1, 0)


def test_kernel_variants_agree():
    """A/B of alternative kernels for the same op: tiled vs untiled weight gradient, conv_s1 vs generic conv, MFMA vs VALU pointwise /
    patch-expand paths.  They must agree to fp32 round-off (all are k-ordered fp32 FMA chains)."""
    VF = _vf()
    import veloxseg_amd.functional as F_
    d = dev()

    def run(cfg):
        for k, v in cfg.items():
            setattr(F_, k, v)
        torch.manual_seed(0)
        res = []
        for (Cin, Cout, K, G, ps, sp) in [(16, 128, 3, 1, 4, (6, 5, 8)), (16, 16, 5, 4, 1, (8, 8, 8)), (32, 32, 3, 4, 1, (4, 6, 8)), (64, 32, 1, 1, 1, (4, 4, 4)),
                                          (16, 16, 1, 4, 1, (8, 8, 8)), (32, 32, 1, 4, 1, (6, 5, 8)), (128, 128, 1, 8, 1, (4, 4, 4)),
                                          (16, 64, 3, 1, 4, (4, 8, 16)), (16, 128, 3, 1, 4, (8, 4, 32)),      # these two hit the LDS-tiled expand kernel ...
                                          (16, 128, 3, 1, 4, (4, 4, 24)), (16, 64, 3, 1, 4, (4, 8, 8))]:       # ... and these with a partly idle last tile (24-wide rows = the shipped 96^3 patches)
            x = rnd(2, Cin, *sp, seed=Cin).to(d).requires_grad_(True)
            w = (rnd(Cout, Cin // G, K, K, K, seed=K) * 0.1).to(d).requires_grad_(True)
            b = rnd(Cout, seed=3).to(d).requires_grad_(True)
            y = VF.conv3d(x, w, b, padding=K // 2, groups=G, pixel_shuffle=ps)
            y.backward(rnd(*y.shape, seed=5).to(d))
            res += [y.detach(), x.grad, w.grad, b.grad]
        torch.cuda.synchronize()
        return res

    base = dict(WGRAD_ENTRY="vx_conv3d_bwd_weight_tiled", USE_S1=True, USE_EXPAND_MFMA=True, PW_MFMA_MAX_V=4096, USE_GCONV1=True, USE_WGRAD_WS=True)
    saved = {k: getattr(F_, k) for k in base}            # (restored below: a flag left changed sends every later test of the session down the python operator bodies)
    ref = run(dict(WGRAD_ENTRY="vx_conv3d_bwd_weight", USE_S1=False, USE_EXPAND_MFMA=False, PW_MFMA_MAX_V=0, USE_GCONV1=False, USE_WGRAD_WS=False))
    try:
        for variant in (base, dict(base, USE_EXPAND_MFMA=False), dict(base, PW_MFMA_MAX_V=0), dict(base, USE_GCONV1=False), dict(base, USE_WGRAD_WS=False)):
            got = run(variant)
            for i, (a, b_) in enumerate(zip(got, ref)):
                close(a, b_, 2e-5 * max(1.0, float(b_.abs().max())), 1e-4, f"variant {variant} tensor {i}")
        from veloxseg_amd import _hip as H
        for knob in ("vx_pw_mfma_set_wide", "vx_expand_set_lds", "vx_expand_set_fwd_wlds"):          # library-side A/B knobs: narrow MFMA tiles / un-tiled expand gradient
            for val in ((0, 2) if knob == "vx_expand_set_lds" else (0,)):
                H.call(knob, val)
                try:
                    got = run(base)
                finally:
                    H.call(knob, 1)
                for i, (a, b_) in enumerate(zip(got, ref)):
                    close(a, b_, 2e-5 * max(1.0, float(b_.abs().max())), 1e-4, f"{knob}={val} tensor {i}")
    finally:
        for k, v in saved.items():
            setattr(F_, k, v)


@pytest.mark.parametrize("Cin,Cout,K,sp", [(1, 16, 4, (16, 16, 8)), (2, 8, 2, (8, 12, 16)), (1, 16, 4, (32, 32, 32))])
def test_patch_conv_via_patchify_matches_direct_conv_and_oracle(Cin, Cout, K, sp):
    """PatchEmbed (kernel == stride): patchify + 1x1 conv (MFMA / VALU by volume) vs the generic direct conv vs aten on the CPU"""
    VF = _vf()
    import veloxseg_amd.functional as F_
    d = dev()
    x = rnd(2, Cin, *sp, seed=1)
    w0 = rnd(Cout, Cin, K, K, K, seed=2) * 0.2
    b0 = rnd(Cout, seed=3)
    gy = None
    res = {}
    try:
        for flag in (True, False):
            F_.USE_PATCHIFY = flag
            w = w0.clone().to(d).requires_grad_(True)
            b = b0.clone().to(d).requires_grad_(True)
            y = VF.conv3d(x.to(d), w, b, stride=K, padding=0)
            gy = rnd(*y.shape, seed=4) if gy is None else gy
            y.backward(gy.to(d))
            torch.cuda.synchronize()
            res[flag] = (y.detach().cpu(), w.grad.cpu(), b.grad.cpu())
    finally:
        F_.USE_PATCHIFY = True
    wc, bc = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    yc = F.conv3d(x, wc, bc, stride=K)
    yc.backward(gy)
    for flag in (True, False):
        close(res[flag][0], yc, 2e-5, 2e-4, f"patchify={flag} y")
        close(res[flag][1], wc.grad, 1e-4 * float(wc.grad.abs().max()), 5e-4, f"patchify={flag} dw")
        close(res[flag][2], bc.grad, 1e-4 * float(bc.grad.abs().max()), 5e-4, f"patchify={flag} db")


@pytest.mark.parametrize("B,Cin,Ctot,c_off,Cout,grid", [(2, 1, 2, 1, 16, (32, 32, 16)), (2, 4, 4, 0, 32, (8, 12, 24)), (1, 1, 1, 0, 16, (16, 16, 4)), (3, 2, 5, 2, 48, (4, 6, 8))],
                         ids=["modality_slice_128x128x64", "brats_4ch_Wo24", "hecktor_like_Wo4", "slice_2_of_5"])
def test_patch_embed_in_place_equals_patchify_plus_pointwise(B, Cin, Ctot, c_off, Cout, grid):
    """vx_patch_embed_fwd / _bwd_weight (no patchified copy) vs vx_patchify_bs + vx_pw_conv_fwd / vx_pw_conv_bwd_weight on a channel slice of a wider tensor:
    the product bit-identical (same order of the sums), the weight / bias gradients to fp32 rounding, and against aten on the CPU"""
    from veloxseg_amd import _hip as H
    d = dev()
    Do, Ho, Wo = grid
    Vo = Do * Ho * Wo
    xw = rnd(B, Ctot, 4 * Do, 4 * Ho, 4 * Wo, seed=1)
    w = rnd(Cout, Cin, 4, 4, 4, seed=2) * 0.2
    bias = rnd(Cout, seed=3)
    gy = rnd(B, Cout, Do, Ho, Wo, seed=4)
    xd, wd, bd, gd = xw.to(d), w.to(d), bias.to(d), gy.to(d)
    xs = xd[:, c_off:c_off + Cin]
    st = H.stream_ptr()
    assert H.query("vx_patch_embed_ok", Cin, Cout, 32, 32, 32, 4) == 1 and H.query("vx_patch_embed_ok", Cin, Cout, 32, 32, 30, 4) == 0
    # reference: the patchified copy and the 1x1 kernels
    ck = Cin * 64
    pat = torch.empty(B, ck, Vo, device=d)
    H.call("vx_patchify_bs", xs.data_ptr(), xs.stride(0), H.P(pat), B, Cin, Do, Ho, Wo, 4, st)
    y_ref = torch.empty(B, Cout, Vo, device=d)
    H.call("vx_pw_conv_fwd", H.P(pat), None, ck, H.P(wd), H.P(bd), H.P(y_ref), B, ck, Cout, Vo, st)
    dw_ref, db_ref = torch.zeros(Cout, ck, device=d), torch.zeros(Cout, device=d)
    H.call("vx_pw_conv_bwd_weight", H.P(pat), None, ck, H.P(gd), H.P(dw_ref), H.P(db_ref), B, ck, Cout, Vo, st)
    y = torch.empty(B, Cout, Vo, device=d)
    H.call("vx_patch_embed_fwd", xs.data_ptr(), xs.stride(0), H.P(wd), H.P(bd), H.P(y), B, Cin, Cout, Do, Ho, Wo, st)
    dw, db = torch.zeros(Cout, ck, device=d), torch.zeros(Cout, device=d)
    H.call("vx_patch_embed_bwd_weight", xs.data_ptr(), xs.stride(0), H.P(gd), H.P(dw), H.P(db), B, Cin, Cout, Do, Ho, Wo, st)
    H.call("vx_patch_embed_bwd_weight", xs.data_ptr(), xs.stride(0), H.P(gd), H.P(dw), None, B, Cin, Cout, Do, Ho, Wo, st)       # accumulates; null bias gradient
    torch.cuda.synchronize()
    if Vo >= 16384 and ck <= 64:        # (the 1x1 product on vx_pw_fwd_v4_k: the same order of the sums)
        assert torch.equal(y, y_ref), float((y - y_ref).abs().max())
    close(y, y_ref, 2e-6 * float(y_ref.abs().max()), 2e-6, "y vs patchified")
    close(dw.cpu() * 0.5, dw_ref.cpu(), 2e-5 * float(dw_ref.abs().max()), 2e-4, "dw vs patchified")
    close(db.cpu(), db_ref.cpu(), 2e-5 * float(db_ref.abs().max()), 2e-4, "db vs patchified")
    wc, bc = w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yc = F.conv3d(xw[:, c_off:c_off + Cin], wc, bc, stride=4)
    yc.backward(gy)
    close(y.cpu().view_as(yc), yc.detach(), 2e-5, 2e-4, "y vs aten")
    close(dw.cpu().view_as(wc.grad) * 0.5, wc.grad, 1e-4 * float(wc.grad.abs().max()), 5e-4, "dw vs aten")
    close(db.cpu(), bc.grad, 1e-4 * float(bc.grad.abs().max()), 5e-4, "db vs aten")


@pytest.mark.parametrize("B,Cin,Cout,V,C1", [(2, 32, 48, 512, 0), (1, 16, 64, 4096, 0), (2, 24, 20, 64, 8), (2, 32, 16, 216, 0)],
                         ids=["V512", "V4096_ksplit", "concat_V64", "V216"])
def test_pw_conv_bwd_fused_equals_two_launches(B, Cin, Cout, V, C1):
    """vx_pw_conv_bwd_fused == vx_pw_conv_mfma(transpose) + vx_pw_conv_bwd_weight, incl. accumulate mode, concat inputs and a null bias gradient"""
    from veloxseg_amd import _hip as H
    d = dev()
    dy, w = rnd(B, Cout, V, seed=1).to(d), rnd(Cout, Cin, seed=2).to(d)
    c1 = C1 or Cin
    x, x2 = rnd(B, c1, V, seed=3).to(d), (rnd(B, Cin - c1, V, seed=4).to(d) if C1 else None)
    st = H.stream_ptr()
    for acc in (0, 1):
        ref_dx, ref_dx2 = torch.full((B, c1, V), 0.5, device=d), (torch.full((B, Cin - c1, V), 0.25, device=d) if C1 else None)
        got_dx, got_dx2 = ref_dx.clone(), (ref_dx2.clone() if C1 else None)
        ref_dw, ref_db, got_dw, got_db = (torch.zeros(Cout, Cin, device=d), torch.zeros(Cout, device=d), torch.zeros(Cout, Cin, device=d), torch.zeros(Cout, device=d))
        H.call("vx_pw_conv_mfma", H.P(dy), None, 0, H.P(w), 1, None, H.P(ref_dx), H.P(ref_dx2), c1, B, Cin, Cout, Cin, V, acc, st)
        H.call("vx_pw_conv_bwd_weight", H.P(x), H.P(x2), c1, H.P(dy), H.P(ref_dw), H.P(ref_db), B, Cin, Cout, V, st)
        H.call("vx_pw_conv_bwd_fused", H.P(dy), H.P(w), H.P(x), H.P(x2), c1, H.P(got_dx), H.P(got_dx2), H.P(got_dw), H.P(got_db) if acc == 0 else None,
               B, Cin, Cout, V, acc, st)
        torch.cuda.synchronize()
        assert torch.equal(got_dx, ref_dx) and (not C1 or torch.equal(got_dx2, ref_dx2))
        close(got_dw, ref_dw, 1e-5 * max(1.0, float(ref_dw.abs().max())), 1e-5, "dw")            # float atomics: summation order only
        if acc == 0:
            close(got_db, ref_db, 1e-5 * max(1.0, float(ref_db.abs().max())), 1e-5, "db")
        else:
            assert float(got_db.abs().max()) == 0.0


@pytest.mark.parametrize("grid,big,heads,c", [([16, 16, 16], [8, 8, 8], 2, 8), ([16, 16, 16], [4, 4, 4], 1, 16), ([8, 8, 8], [4, 4, 4], 4, 32),
                                              ([32, 32, 32], [4, 4, 4], 1, 4), ([32, 32, 32], [4, 4, 4], 2, 8)])
def test_scatter_adjoint_transpose_kernel_equals_general_adjoint(grid, big, heads, c):
    """the specialised adjoints of window_scattering_3d -- LDS transpose (1x1x1 small windows), parallel gather (cells up to 4^3), separable
    reductions (wider cells: the coarse scale of the 32^3 level) -- write exactly what the general (LDS-atomic) adjoint accumulates"""
    from veloxseg_amd import _hip as H
    d = dev()
    pl = O.plan_pwa(grid, big, [1, 1, 1], 2, heads, c, heads * c * 2)
    plan = H.make_plan(grid, pl["n"], heads, pl["small"], pl["nwin"])
    B, M = 2, 2
    pp = H.ctypes.addressof(plan)
    dout = [rnd(B, plan.nb * heads * c, *grid, seed=60 + m).to(d) for m in range(M)]
    res = []
    try:
        for knob in (1, 0):
            H.call("vx_pwa_scatter_set_ident", knob)
            dtok = torch.zeros(B, heads, plan.Ntot, M * plan.l, c, device=d)
            for m in range(M):
                H.call("vx_pwa_scatter_bwd", H.P(dout[m]), H.P(dtok), pp, c, m, M, B, H.stream_ptr())
            torch.cuda.synchronize()
            res.append(dtok)
    finally:
        H.call("vx_pwa_scatter_set_ident", 1)
    assert float(res[0].abs().max()) > 0
    close(res[0], res[1], 2e-6 * max(1.0, float(res[1].abs().max())), 1e-5, "scatter adjoint")       # the general adjoint sums with float atomics


@pytest.mark.parametrize("B,Cc,sp", [(2, 1, (8, 8, 16)), (1, 2, (4, 8, 8))], ids=["one_channel", "two_channels"])
def test_patch_expand_backward_takes_the_weight_scale_from_its_forward(B, Cc, sp):
    """fp16-piece mode (ns = 22): the input gradient with the forward's scale word (vx_expand_bwd_data_mfma_split_ew) == the one that finds max |w| itself, bit for bit"""
    from veloxseg_amd import _hip as H
    d = dev()
    D, Hh, W = sp
    Cout = 64 * Cc
    x, w, bias = rnd(B, 16, *sp, seed=1).to(d), (rnd(Cout, 16, 3, 3, 3, seed=2) * 0.3).to(d), rnd(Cout, seed=3).to(d)
    dy = rnd(B, Cc, 4 * D, 4 * Hh, 4 * W, seed=4).to(d)
    st = H.stream_ptr()
    nws = max(H.query("vx_expand_split_ws_floats", Cc, 22), Cout * 16 * 27)
    wt_f, wt_b1, wt_b2 = (torch.empty(nws, device=d) for _ in range(3))
    y = torch.empty(B, Cc, 4 * D, 4 * Hh, 4 * W, device=d)
    assert H.query("vx_expand_fwd_mfma_split", H.P(x), H.P(w), H.P(bias), H.P(wt_f), H.P(y), B, Cc, D, Hh, W, 22, st) == 0
    off = H.query("vx_expand_split_ew_offset", Cc)
    assert off + 2 <= nws
    dx1, dx2 = torch.empty_like(x), torch.full_like(x, float("nan"))
    assert H.query("vx_expand_bwd_data_mfma_split", H.P(dy), H.P(w), H.P(wt_b1), H.P(dx1), B, Cc, D, Hh, W, 0, 22, st) == 0
    assert H.query("vx_expand_bwd_data_mfma_split_ew", H.P(dy), H.P(w), H.P(wt_b2), H.P(dx2), B, Cc, D, Hh, W, 0, 22, wt_f.data_ptr() + 4 * off, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(dx1, dx2) and float(dx1.abs().max()) > 0
    # both images built ahead (vx_expand_prep_split22), the matrix kernels alone on them: the same bits again
    wt_pf, wt_pb = torch.full((nws,), float("nan"), device=d), torch.full((nws,), float("nan"), device=d)
    H.call("vx_expand_prep_split22", H.P(w), H.P(wt_pf), H.P(wt_pb), Cc, st)
    y3, dx3 = torch.full_like(y, float("nan")), torch.full_like(x, float("nan"))
    assert H.query("vx_expand_fwd_mfma_split_prepared", H.P(x), H.P(bias), H.P(wt_pf), H.P(y3), B, Cc, D, Hh, W, st) == 0
    assert H.query("vx_expand_bwd_data_mfma_split_prepared", H.P(dy), H.P(wt_pb), wt_pf.data_ptr() + 4 * off, H.P(dx3), B, Cc, D, Hh, W, 0, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(y3, y) and torch.equal(dx3, dx1)


@pytest.mark.parametrize("grid,big,heads,c,M", [([16, 16, 16], [8, 8, 8], 2, 8, 2), ([32, 32, 32], [4, 4, 4], 1, 4, 2), ([8, 8, 8], [4, 4, 4], 4, 32, 1), ([4, 4, 4], [4, 4, 4], 2, 16, 2),
                                                ([24, 24, 24], [3, 3, 3], 1, 4, 2), ([16, 16, 8], [4, 4, 2], 2, 8, 3)],
                         ids=["two_scales", "three_scales_incl_separable", "one_modality", "single_scale", "96_L1", "aniso_three_modalities"])
def test_scatter_adjoint_into_an_unzeroed_destination(grid, big, heads, c, M):
    """vx_pwa_scatter_bwd_all_w (no fill launch in front: the sole-owner kernels assign, the identity-scale kernel zeroes the window ranges of the scales that add with
    atomics) into a buffer full of NaNs == vx_pwa_scatter_bwd_all into a zeroed buffer: bit-identical where only sole-owner scales exist, float-atomic order otherwise"""
    import ctypes
    from veloxseg_amd import _hip as H
    d = dev()
    pl = O.plan_pwa(grid, big, [1, 1, 1], 2, heads, c, heads * c * 2)
    plan = H.make_plan(grid, pl["n"], heads, pl["small"], pl["nwin"])
    B = 2
    pp = H.ctypes.addressof(plan)
    dout = [rnd(B, plan.nb * heads * c, *grid, seed=60 + m).to(d) for m in range(M)]
    ptrs = (ctypes.c_void_p * 4)(*([t.data_ptr() for t in dout] + [None] * (4 - M)))
    ref = torch.zeros(B, heads, plan.Ntot, M * plan.l, c, device=d)
    H.call("vx_pwa_scatter_bwd_all", ctypes.addressof(ptrs), H.P(ref), pp, c, M, B, H.stream_ptr())
    got = torch.full_like(ref, float("nan"))
    assert H.query("vx_pwa_scatter_bwd_all_w", ctypes.addressof(ptrs), H.P(got), pp, c, M, B, H.stream_ptr()) == 0
    # ... and with one more buffer to zero (the attention backward's bias-gradient replicas ride on the same launch)
    got2, extra = torch.full_like(ref, float("nan")), torch.full((4 * 1237,), float("nan"), device=d)
    assert H.query("vx_pwa_scatter_bwd_all_wz", ctypes.addressof(ptrs), H.P(got2), pp, c, M, B, H.P(extra), extra.numel(), H.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert float(extra.abs().max()) == 0.0
    close(got2, got, 2e-6 * max(1.0, float(ref.abs().max())), 1e-5, "with / without the extra buffer")
    if plan.nb <= 2:
        assert torch.equal(got2, got)
    assert torch.isfinite(got).all()
    close(got, ref, 2e-6 * max(1.0, float(ref.abs().max())), 1e-5, "scatter adjoint, unzeroed destination")
    if plan.nb <= 2:
        assert torch.equal(got, ref)


@pytest.mark.parametrize("B,Cin,sp", [(2, 2, (64, 64, 64)), (1, 1, (32, 64, 128)), (2, 4, (16, 32, 64)), (4, 2, (128, 128, 128)), (2, 2, (32, 48, 96)), (1, 2, (96, 96, 96))],
                         ids=["m2_64", "m1_aniso", "m4", "bench_shape", "w96_rows_of_24", "shipped_96"])
def test_stem_weight_gradient_mfma_equals_tiled_kernel(B, Cin, sp):
    """Conv3d(k7, s4, p3) weight + bias gradient: MFMA tiles + partial-sum slices vs the tiled VALU kernel (float atomics) vs aten on a small case"""
    from veloxseg_amd import _hip as H
    d = dev()
    Cout = 16
    x = rnd(B, Cin, *sp, seed=1).to(d)
    Do, Ho, Wo = [(n + 6 - 7) // 4 + 1 for n in sp]
    dy = rnd(B, Cout, Do, Ho, Wo, seed=2).to(d)
    st = H.stream_ptr()
    nws = H.query("vx_down_wgrad_ws_floats", B, Cin, *sp, Cout)
    assert nws > 0
    ws = torch.empty(nws, device=d)
    dw, db = torch.zeros(Cout, Cin, 7, 7, 7, device=d), torch.zeros(Cout, device=d)
    H.call("vx_down_wgrad_mfma", H.P(x), H.P(dy), H.P(dw), H.P(db), H.P(ws), nws, B, Cin, *sp, Cout, st)
    rw, rb = torch.zeros_like(dw), torch.zeros_like(db)
    H.call("vx_conv3d_bwd_weight_tiled", H.P(x), None, Cin, H.P(dy), H.P(rw), H.P(rb), B, Cin, *sp, Cout, 7, 4, 3, 1, 1, st)
    torch.cuda.synchronize()
    close(dw, rw, 2e-5 * max(1.0, float(rw.abs().max())), 1e-4, "dw")
    close(db, rb, 2e-5 * max(1.0, float(rb.abs().max())), 1e-4, "db")
    if x.numel() <= 2 * 2 * 64 ** 3:
        xc, wc = x.cpu().double(), torch.zeros(Cout, Cin, 7, 7, 7, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv3d(xc, wc, stride=4, padding=3).backward(dy.cpu().double())
        close(dw.cpu(), wc.grad.float(), 2e-5 * max(1.0, float(wc.grad.abs().max())), 1e-4, "dw vs aten")
    # shapes the tile geometry does not cover (output rows not a multiple of 4 in H): the query says 0 and the entry declines without launching
    assert H.query("vx_down_wgrad_ws_floats", 1, 2, 96, 88, 96, 16) == 0
    assert H.query("vx_down_wgrad_mfma", H.P(x), H.P(dy), H.P(dw), H.P(db), H.P(ws), nws, 1, 2, 96, 88, 96, 16, st) == 1


@pytest.mark.parametrize("nk,act,V,BC", [(3, 1, 32768, 8), (1, 0, 8192, 6), (2, 1, 5000, 3)])
def test_instance_norm_long_rows_two_launch_path_equals_separate_launches(nk, act, V, BC):
    """vx_in_fwd_split / vx_in_bwd_split (partials of all inputs in one launch, consumers fold them) vs vx_in_stats + vx_in_apply_fwd / vx_in_bwd"""
    from veloxseg_amd import _hip as H
    d = dev()
    ys = [(rnd(BC, V, seed=10 + k) * (1.0 + k) + 0.3 * k).to(d) for k in range(nk)]
    res, dout = rnd(BC, V, seed=5).to(d), rnd(BC, V, seed=6).to(d)
    st = H.stream_ptr()
    pad = [None] * (3 - nk)
    # separate launches
    stats = [torch.empty(BC * 2, device=d) for _ in range(nk)]
    for y, s in zip(ys, stats):
        part = torch.empty(BC * 32, device=d, dtype=torch.float64)
        H.call("vx_in_stats", H.P(y), H.P(s), H.P(part, torch.float64), BC, V, 1e-5, st)
    ref = torch.empty(BC, V, device=d)
    H.call("vx_in_apply_fwd", *([H.P(y) for y in ys] + pad), *([H.P(s) for s in stats] + pad), nk, act, H.P(res), H.P(ref), BC, V, st)
    rgrads = []
    for y, s in zip(ys, stats):
        g, ws, part = torch.empty(BC, V, device=d), torch.empty(BC * 2, device=d), torch.empty(BC * 32, device=d, dtype=torch.float64)
        H.call("vx_in_bwd", H.P(dout), H.P(y), H.P(s), act, H.P(ws), H.P(part, torch.float64), H.P(g), BC, V, st)
        rgrads.append(g)
    # two launches per direction
    stats2 = [torch.empty(BC * 2, device=d) for _ in range(nk)]
    part = torch.empty(nk * BC * 32, device=d, dtype=torch.float64)
    out = torch.empty(BC, V, device=d)
    H.call("vx_in_fwd_split", *([H.P(y) for y in ys] + pad), *([H.P(s) for s in stats2] + pad), H.P(part, torch.float64), nk, act, H.P(res), H.P(out), BC, V, 1e-5, st)
    grads = [torch.empty(BC, V, device=d) if k != 1 else None for k in range(nk)]          # input 1 (when present) needs no gradient
    H.call("vx_in_bwd_split", H.P(dout), *([H.P(y) for y in ys] + pad), *([H.P(s) for s in stats2] + pad), H.P(part, torch.float64), nk, act,
           *([H.P(g) for g in grads] + pad), None, BC, V, st)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    for s2, s in zip(stats2, stats):
        assert torch.equal(s2, s)
    for k in range(nk):
        if grads[k] is not None:
            assert torch.equal(grads[k], rgrads[k]), k


@pytest.mark.parametrize("B,Cin,Cout,V,C1", [(2, 16, 48, 32768, 0), (1, 48, 16, 8192, 0), (2, 24, 20, 5000, 8)], ids=["V32768", "V8192", "concat_unaligned"])
def test_pw_conv_bwd_fused_big_equals_two_launches(B, Cin, Cout, V, C1):
    """vx_pw_conv_bwd_fused_big == vx_pw_conv_bwd_data + vx_pw_conv_bwd_weight (large volumes), incl. accumulate mode and concat inputs"""
    from veloxseg_amd import _hip as H
    d = dev()
    dy, w = rnd(B, Cout, V, seed=1).to(d), rnd(Cout, Cin, seed=2).to(d)
    c1 = C1 or Cin
    x, x2 = rnd(B, c1, V, seed=3).to(d), (rnd(B, Cin - c1, V, seed=4).to(d) if C1 else None)
    st = H.stream_ptr()
    for acc in (0, 1):
        ref_dx, ref_dx2 = torch.full((B, c1, V), 0.5, device=d), (torch.full((B, Cin - c1, V), 0.25, device=d) if C1 else None)
        got_dx, got_dx2 = ref_dx.clone(), (ref_dx2.clone() if C1 else None)
        ref_dw, ref_db, got_dw, got_db = (torch.zeros(Cout, Cin, device=d), torch.zeros(Cout, device=d), torch.zeros(Cout, Cin, device=d), torch.zeros(Cout, device=d))
        H.call("vx_pw_conv_bwd_data", H.P(dy), H.P(w), H.P(ref_dx), H.P(ref_dx2), c1, B, Cin, Cout, V, acc, st)
        H.call("vx_pw_conv_bwd_weight", H.P(x), H.P(x2), c1, H.P(dy), H.P(ref_dw), H.P(ref_db), B, Cin, Cout, V, st)
        H.call("vx_pw_conv_bwd_fused_big", H.P(dy), H.P(w), H.P(x), H.P(x2), c1, H.P(got_dx), H.P(got_dx2), H.P(got_dw), H.P(got_db), B, Cin, Cout, V, acc, st)
        torch.cuda.synchronize()
        assert torch.equal(got_dx, ref_dx) and (not C1 or torch.equal(got_dx2, ref_dx2))
        close(got_dw, ref_dw, 1e-5 * max(1.0, float(ref_dw.abs().max())), 1e-5, "dw")
        close(got_db, ref_db, 1e-5 * max(1.0, float(ref_db.abs().max())), 1e-5, "db")


def test_loss_backward_one_launch_for_all_heads_equals_per_head_launches():
    VF = _vf()
    d = dev()
    B, S, ncls, M = 2, 12, 3, 2
    g = torch.Generator().manual_seed(5)
    base = [rnd(B, ncls, S, S, S, seed=i) for i in range(4)] + [rnd(B, M, S, S, S, seed=7)] + [rnd(B, 16, 16, seed=8 + i, scale=0.1) for i in range(1 + M)]
    lab = torch.randint(0, ncls, (B, 1, S, S, S), generator=g).to(d)
    sr = rnd(B, M, S, S, S, seed=99).to(d)
    res = []
    try:
        for flag in (True, False):
            VF.USE_LOSS_BWD4 = flag
            t = [b.clone().to(d).requires_grad_(True) for b in base]
            VF.veloxseg_loss(t, lab, sr, [0.4, 0.3, 0.2, 0.1], 0.5, 2.0, M).backward()
            torch.cuda.synchronize()
            res.append([x.grad.clone() for x in t])
    finally:
        VF.USE_LOSS_BWD4 = True
    for i, (a, b) in enumerate(zip(*res)):          # the 4-voxel kernel contracts / rounds the soft-max normalisation differently: a few ulp
        close(a, b, 2e-6 * max(1e-3, float(b.abs().max())), 1e-5, f"grad {i}")


@pytest.mark.parametrize("grid,big,heads,mdh,C,M", [([16, 16, 16], [8, 8, 8], 2, 8, 32, 2), ([8, 8, 8], [4, 4, 4], 2, 8, 64, 2), ([4, 4, 4], [4, 4, 4], 4, 16, 128, 2),
                                                    ([16, 16, 16], [4, 4, 4], 1, 4, 16, 2), ([32, 32, 32], [4, 4, 4], 1, 4, 16, 1), ([16, 16, 16], [8, 8, 8], 2, 8, 32, 1),
                                                    ([12, 12, 12], [6, 6, 6], 2, 8, 32, 2), ([6, 6, 6], [3, 3, 3], 2, 8, 64, 2), ([3, 3, 3], [3, 3, 3], 4, 16, 128, 2),
                                                    ([24, 24, 24], [3, 3, 3], 1, 4, 16, 2), ([8, 8, 4], [4, 4, 2], 2, 8, 64, 2),
                                                    # (round 6) the BraTS coarse levels: one modality, a handful of 64-token windows -> the one-pass kernel with ONE query tile per block
                                                    ([8, 8, 8], [4, 4, 4], 2, 8, 64, 1), ([4, 4, 4], [4, 4, 4], 4, 16, 128, 1)],
                         ids=["c8v8_ML1024", "c8v16", "c16v32", "c4v8", "c4v4_M1", "c8v8_M1", "96_L2_l216", "96_L3_l27", "96_L4_l27", "96_L1_l27", "aniso_l32", "c8v16_M1_few", "c16v32_M1_few"])
def test_pwa_attention_mfma_kernels_equal_the_valu_kernels(grid, big, heads, mdh, C, M):
    """The MFMA attention kernels (csrc/pwa_mfma.hip) against the fp32-VALU kernels of the same library on the same inputs, dropout ON (p = 0.2: both
    draw the same Philox words for the same (query, key) element): outputs, dq / dk / dv and the bias-table gradient agree to fp32 summation noise.
    Mask 11 = MFMA forward (where l % 64 == 0) + the ONE-pass MFMA backward for every geometry (any l: padded 16-token tiles, incl. the 27- / 216- / 32-token
    windows of the shipped 96^3 and anisotropic configurations; the default mask 3 selects it only where it is the faster kernel: l % 16 != 0); mask 5 = the
    older two-kernel MFMA backward; mask 0 = the VALU kernels.  With l % 4 != 0 the one-pass backward reads the forward's keep bits instead of Philox words."""
    VF = _vf()
    from veloxseg_amd import _hip as H
    d = dev()
    pl = O.plan_pwa(grid, big, [1, 1, 1], 2, heads, mdh, C)
    plan = H.make_plan(grid, pl["n"], heads, pl["small"], pl["nwin"])
    H.call("vx_pwa_attn_set_mfma", 11)
    l = pl["n"][0] * pl["n"][1] * pl["n"][2]
    assert H.query("vx_pwa_attn_bwd1_ok", H.ctypes.addressof(plan), 2, M, pl["c_qk"], pl["c_v"]) == 1, "geometry is expected to take the one-pass MFMA backward"
    assert H.query("vx_pwa_attn_mfma_ok", H.ctypes.addressof(plan), 2, M, pl["c_qk"], pl["c_v"]) == (1 if l % 64 == 0 else 0)
    masks = (11, 5, 0) if l % 64 == 0 else (11, 0)
    n = pl["n"]
    base = []
    for m in range(M):
        base += [rnd(2, pl["ch_qk"], *grid, seed=10 + m), rnd(2, pl["ch_qk"], *grid, seed=20 + m), rnd(2, pl["ch_v"], *grid, seed=30 + m)]
    res = {}
    try:
        for on in masks:
            H.call("vx_pwa_attn_set_mfma", on)
            VF.manual_seed(77, d)
            table = (rnd((2 * n[0] - 1) * (2 * n[1] - 1) * (2 * n[2] - 1), heads, seed=4, scale=0.5)).to(d).requires_grad_(True)
            t = [b.clone().to(d).requires_grad_(True) for b in base]
            outs = VF.pwa_core(table, plan, pl["c_qk"], pl["c_v"], t, p_attn=0.2, site=9)
            gouts = [rnd(*o.shape, seed=50 + i).to(d) for i, o in enumerate(outs)]
            torch.autograd.backward(outs, gouts)
            torch.cuda.synchronize()
            res[on] = [o.detach() for o in outs] + [x.grad for x in t] + [table.grad.clone()]
    finally:
        H.call("vx_pwa_attn_set_mfma", 3)
    for on in masks[:-1]:
        for i, (a, b) in enumerate(zip(res[on], res[0])):
            close(a, b, 3e-5 * max(1.0, float(b.abs().max())), 2e-4, f"mfma (mask {on}) vs valu tensor {i}")


@pytest.mark.parametrize("p_attn", [0.2, 0.0], ids=["dropout", "no_dropout"])
@pytest.mark.parametrize("scale", [1.0, 300.0, 1e-4], ids=["unit", "large", "tiny"])
@pytest.mark.parametrize("grid,big,heads,mdh,C,B", [([16, 16, 16], [8, 8, 8], 2, 8, 32, 2), ([16, 16, 16], [4, 4, 4], 1, 4, 12, 2), ([8, 8, 8], [8, 8, 8], 2, 8, 8, 1),
                                                    ([8, 8, 8], [4, 4, 4], 1, 4, 8, 3),
                                                    # (round 6) ragged windows = the shipped geometries: 27 / 216 tokens (96^3 patches, windows [3, 6, ..]), 32 (Hecktor), 125
                                                    ([12, 12, 12], [3, 3, 3], 1, 4, 12, 2), ([12, 12, 12], [6, 6, 6], 2, 8, 32, 2), ([8, 8, 4], [4, 4, 2], 1, 4, 8, 2),
                                                    ([10, 10, 10], [5, 5, 5], 2, 8, 16, 1), ([6, 6, 6], [6, 6, 6], 1, 4, 4, 3)],
                         ids=["L2_512tok_c8", "L1_64tok_c4", "L2_one_window_per_head", "L1_small_grid", "ragged_27tok_c4", "ragged_216tok_c8", "ragged_32tok_aniso_c4", "ragged_125tok_c8",
                              "ragged_216tok_c4_one_window"])
def test_pwa_attention_f16_pipe_backward_equals_the_fp32_kernels(grid, big, heads, mdh, C, B, scale, p_attn):
    """The one-pass attention backward on the 16x16x32 f16 matrix pipe (csrc/pwa_mfma.hip vx_pwa_attn_bwd1h_k: levels 1 / 2 of the 128^3 configurations -- windows of
    64 / 512 tokens, two modalities, head widths 4 / 8) against the fp32-VALU kernels of the same library on the same inputs and the same dropout mask (the f16
    kernel reads the keep bits the forward stored, the VALU kernels draw the same Philox words): dq / dk / dv and the bias-table gradient agree to fp32 summation
    noise.  Every operand enters the MFMAs as two fp16 pieces of the value scaled by a power of two from the block's maxima: the `large` / `tiny` cases put the
    incoming gradient (x scale) and the values (x 30 / x 0.03) far outside fp16's comfortable range -- the scaling has to bring them back."""
    VF = _vf()
    from veloxseg_amd import _hip as H
    d = dev()
    M = 2
    pl = O.plan_pwa(grid, big, [1, 1, 1], 2, heads, mdh, C)
    plan = H.make_plan(grid, pl["n"], heads, pl["small"], pl["nwin"])
    pp = H.ctypes.addressof(plan)
    H.call("vx_pwa_attn_set_f16_bwd_ragged", 2)                # (every ragged length: by default the short windows -- 27 / 32 tokens -- stay on the fp32 kernels, which are faster there)
    try:
        assert H.query("vx_pwa_attn_bwd1h_ok", pp, B, M, pl["c_qk"], pl["c_v"]) == 1 and H.query("vx_pwa_attn_mbits_useful", pp, B, M, pl["c_qk"], pl["c_v"]) == 1
        _f16_pipe_backward_case(VF, H, d, pl, plan, B, M, scale, p_attn)
    finally:
        H.call("vx_pwa_attn_set_f16_bwd_ragged", 1)
    l = pl["n"][0] * pl["n"][1] * pl["n"][2]
    assert H.query("vx_pwa_attn_bwd1h_ok", pp, B, M, pl["c_qk"], pl["c_v"]) == (1 if (l % 64 == 0 or 4 * l >= 3 * ((l + 63) // 64 * 64)) else 0)


def _f16_pipe_backward_case(VF, H, d, pl, plan, B, M, scale, p_attn):
    grid = pl["grid"]
    heads = pl["heads"]
    n = pl["n"]
    vs = 30.0 if scale > 1 else (0.03 if scale < 1 else 1.0)
    base = []
    for m in range(M):
        base += [rnd(B, pl["ch_qk"], *grid, seed=10 + m), rnd(B, pl["ch_qk"], *grid, seed=20 + m), rnd(B, pl["ch_v"], *grid, seed=30 + m) * vs]
    res = {}
    try:
        for f16 in (1, 0):
            H.call("vx_pwa_attn_set_f16_bwd", f16)
            VF.manual_seed(77, d)
            table = (rnd((2 * n[0] - 1) * (2 * n[1] - 1) * (2 * n[2] - 1), heads, seed=4, scale=0.5)).to(d).requires_grad_(True)
            t = [b.clone().to(d).requires_grad_(True) for b in base]
            outs = VF.pwa_core(table, plan, pl["c_qk"], pl["c_v"], t, p_attn=p_attn, site=9)
            gouts = [rnd(*o.shape, seed=50 + i).to(d) * scale for i, o in enumerate(outs)]
            torch.autograd.backward(outs, gouts)
            torch.cuda.synchronize()
            res[f16] = [o.detach() for o in outs] + [x.grad for x in t] + [table.grad.clone()]
    finally:
        H.call("vx_pwa_attn_set_f16_bwd", 1)
    for i, (a, b) in enumerate(zip(res[1], res[0])):
        assert torch.isfinite(a).all()
        close(a, b, 1e-5 * float(b.abs().max()), 2e-4, f"f16 pipe vs fp32 kernels, tensor {i}")


@pytest.mark.parametrize("p_attn", [0.2, 0.0], ids=["dropout", "no_dropout"])
@pytest.mark.parametrize("scale", [1.0, 300.0], ids=["unit", "large"])
@pytest.mark.parametrize("grid,big,heads,mdh,C,B", [([16, 16, 16], [4, 4, 4], 1, 4, 12, 2), ([16, 16, 16], [8, 8, 8], 2, 8, 32, 2), ([12, 12, 12], [6, 6, 6], 2, 8, 32, 2), ([8, 8, 8], [8, 8, 8], 1, 4, 4, 1)],
                         ids=["brats_L1_64tok_c4", "brats_L2_512tok_c8", "brats96_L2_216tok_c8", "one_window_512tok_c4"])
def test_pwa_attention_f16_pipe_backward_one_modality(grid, big, heads, mdh, C, B, scale, p_attn):
    """The same kernel with ONE modality per window (MF = 1 instance; BraTS: `in_ch = [4]`, reference config/models_config_brats2021.json) against the fp32 kernels"""
    VF = _vf()
    from veloxseg_amd import _hip as H
    d = dev()
    M = 1
    pl = O.plan_pwa(grid, big, [1, 1, 1], 2, heads, mdh, C)
    plan = H.make_plan(grid, pl["n"], heads, pl["small"], pl["nwin"])
    pp = H.ctypes.addressof(plan)
    assert pl["c_qk"] == pl["c_v"] and pl["c_qk"] in (4, 8), (pl["c_qk"], pl["c_v"])
    # OFF by default: measured no gain (brats128 B = 2: 36 windows of 512 tokens -- 68 vs 74 us alone at level 2, 38 vs 35 at level 1; the step 746 vs 751 patches/s)
    assert H.query("vx_pwa_attn_bwd1h_ok", pp, B, M, pl["c_qk"], pl["c_v"]) == 0
    H.call("vx_pwa_attn_set_f16_bwd_m1", 1)
    H.call("vx_pwa_attn_set_f16_bwd_ragged", 2)
    try:
        assert H.query("vx_pwa_attn_bwd1h_ok", pp, B, M, pl["c_qk"], pl["c_v"]) == 1 and H.query("vx_pwa_attn_mbits_useful", pp, B, M, pl["c_qk"], pl["c_v"]) == 1
        _f16_pipe_backward_case(VF, H, d, pl, plan, B, M, scale, p_attn)
    finally:
        H.call("vx_pwa_attn_set_f16_bwd_m1", 0)
        H.call("vx_pwa_attn_set_f16_bwd_ragged", 1)


@pytest.mark.parametrize("grid,big,heads,mdh,C,M", [([16, 16, 16], [8, 8, 8], 2, 8, 32, 2), ([16, 16, 16], [4, 4, 4], 1, 4, 16, 2), ([8, 8, 8], [4, 4, 4], 2, 8, 64, 1),
                                                    ([8, 8, 16], [4, 4, 8], 2, 8, 32, 3)],
                         ids=["L2_512tok", "L1_64tok", "M1", "M3_aniso"])
def test_pwa_attention_valu_backward_reads_the_forwards_keep_bits(grid, big, heads, mdh, C, M):
    """With aligned windows (l % 4 == 0: every 128^3 geometry) the fp32-VALU backward reads the dropout keep bits the forward stored (one 16-bit word per
    query and 16-key tile) in BOTH of its passes instead of re-drawing the Philox words: the masks are the same bits, so dq / dk / dv and the bias-table
    gradient equal those of the re-drawing kernels to float-atomic noise; the forward (MFMA or VALU) is unchanged.  (An A/B variant, off by default: the
    backward is bound by LDS latency, not by the Philox arithmetic, and the step time did not move.)"""
    VF = _vf()
    from veloxseg_amd import _hip as H
    d = dev()
    pl = O.plan_pwa(grid, big, [1, 1, 1], 2, heads, mdh, C)
    plan = H.make_plan(grid, pl["n"], heads, pl["small"], pl["nwin"])
    pp = H.ctypes.addressof(plan)
    n = pl["n"]
    base = []
    for m in range(M):
        base += [rnd(2, pl["ch_qk"], *grid, seed=10 + m), rnd(2, pl["ch_qk"], *grid, seed=20 + m), rnd(2, pl["ch_v"], *grid, seed=30 + m)]
    res = {}
    try:
        H.call("vx_pwa_attn_set_f16_bwd", 0)               # (levels 1 / 2 would otherwise take the f16-pipe one-pass backward)
        H.call("vx_pwa_attn_set_short", 0)                 # (round 6: single-modality windows would otherwise take the one-pass MFMA backward)
        for bits in (1, 0):
            H.call("vx_pwa_attn_set_valu_bits", bits)
            assert H.query("vx_pwa_attn_bwd1_ok", pp, 2, M, pl["c_qk"], pl["c_v"]) == 0          # the default rule keeps these geometries on the VALU backward
            assert H.query("vx_pwa_attn_mbits_useful", pp, 2, M, pl["c_qk"], pl["c_v"]) == bits
            VF.manual_seed(77, d)
            table = (rnd((2 * n[0] - 1) * (2 * n[1] - 1) * (2 * n[2] - 1), heads, seed=4, scale=0.5)).to(d).requires_grad_(True)
            t = [b.clone().to(d).requires_grad_(True) for b in base]
            outs = VF.pwa_core(table, plan, pl["c_qk"], pl["c_v"], t, p_attn=0.2, site=9)
            gouts = [rnd(*o.shape, seed=50 + i).to(d) for i, o in enumerate(outs)]
            torch.autograd.backward(outs, gouts)
            torch.cuda.synchronize()
            res[bits] = [o.detach() for o in outs] + [x.grad for x in t] + [table.grad.clone()]
    finally:
        H.call("vx_pwa_attn_set_valu_bits", 0)
        H.call("vx_pwa_attn_set_f16_bwd", 1)
        H.call("vx_pwa_attn_set_short", 1)
    for i, (a, b) in enumerate(zip(res[1], res[0])):
        if i < M:
            assert torch.equal(a, b), f"output {i}: the forward must not depend on whether it stores the keep bits"
        else:
            close(a, b, 2e-5 * max(1.0, float(b.abs().max())), 1e-4, f"keep bits vs re-drawn words, tensor {i}")


@pytest.mark.parametrize("grid,big,heads,mdh,C,M", [([16, 16, 16], [8, 8, 8], 2, 8, 32, 2), ([8, 8, 8], [4, 4, 4], 2, 8, 64, 2), ([4, 4, 4], [4, 4, 4], 4, 16, 128, 2),
                                                    ([16, 16, 16], [4, 4, 4], 1, 4, 16, 2), ([32, 32, 32], [4, 4, 4], 1, 4, 16, 3)],
                         ids=["c8v8", "c8v16", "c16v32", "c4v8", "c4v4_M3"])
def test_pwa_channel_vectorised_gather_equals_the_per_channel_kernels(grid, big, heads, mdh, C, M):
    """vx_pwa_gather_all_fwd / _bwd and vx_pwa_scatter_fwd: the kernels that move 4 or 8 channels per lane (csrc/pwa.hip *_v_k) against the one-lane-per-channel kernels:
    the pooled tokens are the same (outputs equal to an ulp of the scatter's interpolation) and the gradients are routed to the same voxels (both take the first maximum of a cell)."""
    VF = _vf()
    from veloxseg_amd import _hip as H
    d = dev()
    pl = O.plan_pwa(grid, big, [1, 1, 1], 2, heads, mdh, C)
    plan = H.make_plan(grid, pl["n"], heads, pl["small"], pl["nwin"])
    n = pl["n"]
    base = []
    for m in range(M):
        base += [rnd(2, pl["ch_qk"], *grid, seed=10 + m), rnd(2, pl["ch_qk"], *grid, seed=20 + m), rnd(2, pl["ch_v"], *grid, seed=30 + m)]
    base[0][:, :, :2, :2, :2] = 0.25            # ties inside pooling cells: the first maximum must win in both forms
    res = {}
    try:
        for on in (1, 0):
            H.call("vx_pwa_gather_set_vec", on)
            table = (rnd((2 * n[0] - 1) * (2 * n[1] - 1) * (2 * n[2] - 1), heads, seed=4, scale=0.5)).to(d).requires_grad_(True)
            t = [b.clone().to(d).requires_grad_(True) for b in base]
            outs = VF.pwa_core(table, plan, pl["c_qk"], pl["c_v"], t, p_attn=0.0, site=9)
            gouts = [rnd(*o.shape, seed=50 + i).to(d) for i, o in enumerate(outs)]
            torch.autograd.backward(outs, gouts)
            torch.cuda.synchronize()
            res[on] = [o.detach() for o in outs] + [x.grad for x in t] + [table.grad.clone()]
    finally:
        H.call("vx_pwa_gather_set_vec", 1)
    for i, (a, b) in enumerate(zip(res[1], res[0])):
        if i < M:        # same pooled tokens; the scatter forward's lerps are contracted differently by the two kernels (1 ulp)
            close(a, b, 2e-6 * max(1.0, float(b.abs().max())), 2e-6, f"output {i}")
        else:                    # the attention backward in between accumulates with float atomics: equal routing, summation-order noise
            assert torch.equal(a == 0, b == 0), f"gradient {i}: different arg-max routing"
            close(a, b, 1e-4 * max(1.0, float(b.abs().max())), 1e-4, f"gradient {i}")      # (2e-5 of the maximum is seen run to run with EITHER form)


@pytest.mark.parametrize("ncls,B,S,labdtype,factors", [(2, 2, (32, 32, 32), torch.int64, (2, 4, 8)), (4, 1, (32, 48, 64), torch.uint8, (2, 4, 8)), (3, 2, (16, 16, 128), torch.int32, (2, 4, 8)),
                                                       (2, 2, (24, 24, 96), torch.int64, (2, 4, 8)), (4, 1, (16, 24, 48), torch.uint8, (2, 4, 8)),
                                                       (2, 2, (32, 64, 128), torch.uint8, (8, 16, 32)), (3, 1, (16, 40, 64), torch.int64, (4, 8, 16))],
                         ids=["c2", "c4_aniso", "c3", "c2_w96", "c4_w48", "c2_model_ratios", "c3_ragged_rows"])
@pytest.mark.parametrize("columns", [1, 0], ids=["columns", "rows"])
def test_loss_with_fused_deep_supervision_upsampling(ncls, B, S, labdtype, factors, columns):
    """veloxseg_loss on heads that stay on their own grids (csrc/loss_ds.hip interpolates inside the kernels) == up-sample (vx_upsample_trilinear) then
    veloxseg_loss == the oracle (F.interpolate + CE + Dice): loss 1e-5 relative, gradients of every head 1e-4 of their scale.  Both thread maps of the fused kernels:
    column owners (the 128^3 patches with heads at 1/8, 1/16, 1/32 -- `c2_model_ratios` -- and W = 96 / 48, where the last 256 % (W/4) threads of a block idle) and
    the row sweep."""
    VF = _vf()
    from veloxseg_amd import _hip as H
    d = dev()
    H.call("vx_seg_loss_ds_set_columns", columns)
    try:
        _ds_loss_case(VF, d, ncls, B, S, labdtype, factors)
    finally:
        H.call("vx_seg_loss_ds_set_columns", 1)


def _ds_loss_case(VF, d, ncls, B, S, labdtype, factors):
    heads = [rnd(B, ncls, *S, seed=1)] + [rnd(B, ncls, *[max(s // f, 2) for s in S], seed=2 + k) for k, f in enumerate(factors)]
    lab = torch.randint(0, ncls, (B, 1, *S), generator=torch.Generator().manual_seed(3)).to(labdtype)
    w = (0.25, 0.25, 0.25, 0.25)
    res = []
    for fused in (True, False):
        hs = [h.clone().to(d).requires_grad_(True) for h in heads]
        outs = hs if fused else [hs[0]] + [VF.upsample_trilinear(h, S) for h in hs[1:]]
        loss = VF.seg_only_loss(outs, lab.to(d), w)
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss), [h.grad.clone() for h in hs]))
    hc = [h.clone().requires_grad_(True) for h in heads]
    up = [hc[0]] + [F.interpolate(h, size=S, mode="trilinear", align_corners=True) for h in hc[1:]]
    ref = sum(wi * O.seg_loss(u, lab.long()) for wi, u in zip(w, up))
    ref.backward()
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[1][0]), (res[0][0], res[1][0])
    assert abs(res[0][0] - float(ref)) <= 1e-5 * abs(float(ref)), (res[0][0], float(ref))
    for k in range(4):
        g = hc[k].grad
        tol = 1e-4 * float(g.abs().max())
        close(res[0][1][k], g, tol, 1e-4, f"fused: d head {k} vs oracle")
        close(res[0][1][k], res[1][1][k], tol, 1e-4, f"fused vs unfused: d head {k}")


@pytest.mark.parametrize("C,G,K,sp,B", [(16, 4, 5, (32, 32, 32), 2), (16, 4, 3, (32, 32, 32), 2), (32, 4, 5, (16, 16, 16), 2), (32, 4, 3, (16, 16, 16), 3),
                                        (64, 8, 5, (8, 8, 8), 2), (64, 16, 3, (8, 8, 8), 2), (128, 16, 5, (4, 4, 4), 2), (32, 8, 5, (6, 5, 12), 2),
                                        (16, 4, 5, (5, 7, 20), 1)])
def test_row_sliding_weight_gradient_of_the_jlc_convs(C, G, K, sp, B):
    """vx_wgrad_rows_k (a thread owns a kw row of taps, csrc/conv_wgrad.hip) vs the (ci, tap)-pair kernel vs aten in fp64: the JLC grouped convs
    (conv_blocks.py:51-58) at every level of the 128^3 configurations, widths that only take the 4-voxel chunks (W = 12, 20), ragged D / H tiles, and
    the partial-sum (deterministic) variant"""
    from veloxseg_amd import _hip as H
    d = dev()
    x = rnd(B, C, *sp, seed=1).to(d)
    dy = rnd(B, C, *sp, seed=2).to(d)
    Cg = C // G
    ref_w = torch.nn.grad.conv3d_weight(x.double().cpu(), (C, Cg, K, K, K), dy.double().cpu(), padding=K // 2, groups=G)
    ref_b = dy.double().cpu().sum(dim=(0, 2, 3, 4))
    st = H.stream_ptr()
    outs = {}
    for rows in (1, 0):
        H.call("vx_wgrad_set_rows", rows)
        try:
            dw = torch.zeros(C, Cg, K, K, K, device=d)
            db = torch.zeros(C, device=d)
            H.call("vx_conv3d_bwd_weight_tiled", H.P(x), None, C, H.P(dy), H.P(dw), H.P(db), B, C, *sp, C, K, 1, K // 2, G, 1, st)
            nws = H.query("vx_conv3d_bwd_weight_ws_floats", B, C, *sp, C, K, 1, K // 2, G, 1)
            dw2 = torch.zeros_like(dw)
            if nws > 0:
                ws = torch.empty(nws, device=d)
                H.call("vx_conv3d_bwd_weight_tiled_ws", H.P(x), None, C, H.P(dy), H.P(dw2), None, H.P(ws), nws, B, C, *sp, C, K, 1, K // 2, G, 1, st)
            else:
                dw2 = dw.clone()
            torch.cuda.synchronize()
        finally:
            H.call("vx_wgrad_set_rows", 1)
        tol = 2e-5 * max(1.0, float(ref_w.abs().max()))
        close(dw, ref_w, tol, 1e-4, f"rows={rows} dw")
        close(dw2, ref_w, tol, 1e-4, f"rows={rows} dw (partial-sum workspace)")
        close(db, ref_b, 2e-5 * max(1.0, float(ref_b.abs().max())), 1e-4, f"rows={rows} db")
        outs[rows] = dw
    close(outs[1], outs[0], 2e-5 * max(1.0, float(ref_w.abs().max())), 1e-4, "rows vs pairs")


@pytest.mark.parametrize("grid,big,heads,mdh,C,M,B", [([16, 16, 16], [8, 8, 8], 2, 8, 32, 2, 2), ([8, 8, 8], [4, 4, 4], 2, 8, 64, 2, 3), ([16, 16, 16], [4, 4, 4], 1, 4, 16, 2, 2),
                                                      ([6, 6, 6], [3, 3, 3], 2, 8, 32, 2, 1), ([12, 12, 12], [6, 6, 6], 1, 4, 16, 1, 2)],
                         ids=["c8v8_ML1024", "c8v16", "c4v8", "w3_ragged_units", "w6_M1"])
def test_pwa_attention_backward_in_one_launch_equals_the_two_pass_backward(grid, big, heads, mdh, C, M, B):
    """vx_pwa_attn_bwd with the dQ and dK/dV passes interleaved in ONE launch (the dK/dV blocks recompute delta = rowsum(dO * O), the bias-gradient
    replicas are folded by their own kernel) against the two-launch form, dropout ON: dq / dk / dv to fp32 round-off (same arithmetic in the same order), the
    bias-table gradient to float-atomic noise.  The 3^3 / 6^3 windows are the 96^3 configurations (blocks whose units straddle heads keep all bias columns)."""
    VF = _vf()
    from veloxseg_amd import _hip as H
    d = dev()
    pl = O.plan_pwa(grid, big, [1, 1, 1], 2, heads, mdh, C)
    plan = H.make_plan(grid, pl["n"], heads, pl["small"], pl["nwin"])
    n = pl["n"]
    base = []
    for m in range(M):
        base += [rnd(B, pl["ch_qk"], *grid, seed=10 + m), rnd(B, pl["ch_qk"], *grid, seed=20 + m), rnd(B, pl["ch_v"], *grid, seed=30 + m)]
    res = {}
    try:
        for on in (1, 0):
            H.call("vx_pwa_attn_set_fused_bwd", on)
            VF.manual_seed(77, d)
            table = (rnd((2 * n[0] - 1) * (2 * n[1] - 1) * (2 * n[2] - 1), heads, seed=4, scale=0.5)).to(d).requires_grad_(True)
            t = [b.clone().to(d).requires_grad_(True) for b in base]
            outs = VF.pwa_core(table, plan, pl["c_qk"], pl["c_v"], t, p_attn=0.2, site=9)
            gouts = [rnd(*o.shape, seed=50 + i).to(d) for i, o in enumerate(outs)]
            torch.autograd.backward(outs, gouts)
            torch.cuda.synchronize()
            res[on] = [x.grad for x in t] + [table.grad.clone()]
    finally:
        H.call("vx_pwa_attn_set_fused_bwd", 1)
    for i, (a, b) in enumerate(zip(res[1][:-1], res[0][:-1])):      # (the two instantiations of a pass may contract a * b + c differently: not bit-equal)
        close(a, b, 4e-6 * max(1.0, float(b.abs().max())), 1e-5, f"input gradient {i}: one launch vs two")
    close(res[1][-1], res[0][-1], 3e-5 * max(1.0, float(res[0][-1].abs().max())), 2e-4, "bias-table gradient")


@pytest.mark.parametrize("B,C,sp", [(2, 16, (8, 8, 8)), (3, 32, (4, 6, 5)), (2, 128, (4, 4, 4)), (1, 64, (16, 16, 16))])
def test_layernorm_backward_halves_equal_the_whole(B, C, sp):
    """vx_ln_cf_bwd_data + vx_ln_cf_bwd_param (the input gradient now, the parameter gradients whenever the caller likes: csrc/norm.hip) == vx_ln_cf_bwd
    and == vx_ln_cf_bwd_add, bit for bit (the same two kernels, launched by two entries)"""
    from veloxseg_amd import _hip as H
    d = dev()
    V = sp[0] * sp[1] * sp[2]
    x, dout, add = rnd(B, C, *sp, seed=1).to(d), rnd(B, C, *sp, seed=2).to(d), rnd(B, C, *sp, seed=3).to(d)
    gamma = rnd(C, seed=4).to(d)
    st = H.stream_ptr()
    for with_add in (False, True):
        dx0, dx1 = torch.empty_like(x), torch.empty_like(x)
        dg0, db0, dg1, db1 = (torch.zeros(C, device=d) for _ in range(4))
        ws0, ws1 = torch.empty(2 * B * V, device=d), torch.empty(2 * B * V, device=d)
        if with_add:
            H.call("vx_ln_cf_bwd_add", H.P(x), H.P(gamma), H.P(dout), H.P(add), H.P(dx0), H.P(dg0), H.P(db0), H.P(ws0), B, C, V, 1e-6, st)
        else:
            H.call("vx_ln_cf_bwd", H.P(x), H.P(gamma), H.P(dout), H.P(dx0), H.P(dg0), H.P(db0), H.P(ws0), B, C, V, 1e-6, st)
        H.call("vx_ln_cf_bwd_data", H.P(x), H.P(gamma), H.P(dout), H.P(add) if with_add else None, H.P(dx1), H.P(ws1), B, C, V, 1e-6, st)
        H.call("vx_ln_cf_bwd_param", H.P(x), H.P(dout), H.P(ws1), H.P(dg1), H.P(db1), B, C, V, st)
        torch.cuda.synchronize()
        assert torch.equal(dx0, dx1)
        close(dg1, dg0, 1e-5 * max(1.0, float(dg0.abs().max())), 1e-5, "dgamma")      # float atomics across chunks
        close(db1, db0, 1e-5 * max(1.0, float(db0.abs().max())), 1e-5, "dbeta")

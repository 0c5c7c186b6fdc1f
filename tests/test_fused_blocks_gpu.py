"""GPU parity of the fused block kernels (csrc/jlc.hip, csrc/mlp.hip): the JLC block (reference conv_blocks.py:41-75) and the FFN tail
(PWA.py:437 + attention_utils.py:45-71), forward + every gradient,
  (a) against the CPU oracle with dropout off (tolerances stated per test, fp32), and
  (b) against the per-operator kernels of the same library with dropout ON (same Philox masks: outputs equal to summation-order noise)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import veloxseg_oracle as O  # noqa: E402  (checker only)


def _close(a, b, atol, rtol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    if not bool((err <= tol).all()):
        i = int((err - tol).argmax())
        raise AssertionError(f"{what}: max abs err {float(err.max()):.3e} (ref max {float(b.abs().max()):.3e}); worst idx {i}: got {float(a.flatten()[i]):.6e} "
                             f"want {float(b.flatten()[i]):.6e}; bad frac {float((err > tol).double().mean()):.3e}")


def _jlc_module(C, groups, r, p, seed):
    from veloxseg_amd.model.components.conv_blocks import JLC
    from veloxseg_amd import functional as VF
    torch.manual_seed(seed)
    VF.reset_dropout_sites()          # the same dropout stream for every module built by this helper
    m = JLC(C, kernel_sizes=(1, 3, 5), groups=groups, epansion_factor=r, dropout=p)
    with torch.no_grad():
        for q in m.parameters():
            q.copy_(torch.randn_like(q) * (0.3 if q.dim() > 1 else 0.1))
    return m


def _oracle_sd(m, pre="blk."):
    return {pre + k: v.detach().clone().cpu().requires_grad_(True) for k, v in m.state_dict().items()}


JLC_CASES = [
    # name, B, C, groups, expansion, spatial
    ("L1_w4_32cube", 2, 16, 4, 3, (32, 32, 32)),
    ("L2_w8_16cube", 2, 32, 4, 3, (16, 16, 16)),
    ("L1_aniso", 1, 16, 4, 3, (8, 12, 20)),
    ("L2_small", 2, 32, 4, 3, (8, 8, 8)),
    ("w16", 1, 32, 2, 3, (8, 8, 16)),
    # the coarse levels (channel stage on the tile-GEMM kernels of csrc/pwa_fused.hip: vx_inmlp_*)
    ("L3_w8_8cube", 4, 64, 8, 2, (8, 8, 8)),
    ("L4_w16_4cube", 4, 128, 8, 2, (4, 4, 4)),
    ("L3_96_6cube", 2, 64, 8, 2, (6, 6, 6)),
    ("L3_aniso", 1, 64, 8, 2, (8, 8, 4)),
]


@pytest.mark.parametrize("case", JLC_CASES, ids=[c[0] for c in JLC_CASES])
def test_fused_jlc_block_vs_oracle(case):
    """fused path == oracle.jlc (p = 0): output 2e-4 abs / 2e-4 rel, gradients 1e-3 relative to the gradient's scale"""
    from veloxseg_amd import functional as VF
    _, B, C, G, r, sp = case
    cm = VF.cpp_module()
    assert cm is not None
    m = _jlc_module(C, G, r, 0.0, 3).cuda().train()
    x = torch.randn(B, C, *sp, generator=torch.Generator().manual_seed(5))
    xg = x.cuda().requires_grad_(True)
    cm.set_fuse_blocks(True)
    out = m(xg)
    gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(6))
    out.backward(gy.cuda())
    torch.cuda.synchronize()
    sd = _oracle_sd(m)
    xc = x.clone().requires_grad_(True)
    ref = O.jlc(xc, sd, "blk.", G, 0.0, True)
    ref.backward(gy)
    _close(out, ref, 2e-4, 2e-4, "jlc out")
    _close(xg.grad, xc.grad, 1e-3 * float(xc.grad.abs().max()), 1e-3, "jlc dx")
    for k, p in m.named_parameters():
        g_ref = sd["blk." + k].grad
        if "spatial_convs" in k and k.endswith("bias"):
            continue       # behind an InstanceNorm: zero by construction (the reference's value is round-off); not computed
        _close(p.grad, g_ref, 2e-3 * float(g_ref.abs().max()) + 1e-6, 2e-3, f"jlc d{k}")


_DROP_CASES = JLC_CASES[:3] + JLC_CASES[5:7]


@pytest.mark.parametrize("case", _DROP_CASES, ids=[c[0] for c in _DROP_CASES])
def test_fused_jlc_block_equals_per_operator_kernels_with_dropout(case):
    from veloxseg_amd import functional as VF
    _, B, C, G, r, sp = case
    cm = VF.cpp_module()
    res = {}
    try:
        for fused in (True, False):
            cm.set_fuse_blocks(fused)
            m = _jlc_module(C, G, r, 0.1, 3).cuda().train()
            VF.manual_seed(77, "cuda")
            x = torch.randn(B, C, *sp, generator=torch.Generator().manual_seed(5)).cuda().requires_grad_(True)
            out = m(x)
            gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(6)).cuda()
            out.backward(gy)
            torch.cuda.synchronize()
            res[fused] = (out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()})
    finally:
        cm.set_fuse_blocks(True)
    _close(res[True][0], res[False][0], 1e-4, 1e-4, "out")
    _close(res[True][1], res[False][1], 1e-3 * float(res[False][1].abs().max()), 1e-3, "dx")
    for k in res[True][2]:
        if "spatial_convs" in k and k.endswith("bias"):
            continue
        g = res[False][2][k]
        _close(res[True][2][k], g, 2e-3 * float(g.abs().max()) + 1e-6, 2e-3, f"d{k}")


FFN_CASES = [("L1", 2, 16, 3, (16, 16, 32)), ("L2", 2, 32, 3, (8, 16, 16)), ("ragged", 1, 16, 3, (6, 6, 7 * 4))]


def _ffn_modules(C, r, p, seed):
    from veloxseg_amd.model.components.attention_utils import FFN, LayerNorm
    from veloxseg_amd import functional as VF
    torch.manual_seed(seed)
    VF.reset_dropout_sites()
    ffn = FFN(C, expansion_ratio=r, dropout_rate=p)
    ln = LayerNorm(C)
    with torch.no_grad():
        for q in list(ffn.parameters()) + list(ln.parameters()):
            q.copy_(torch.randn_like(q) * 0.3 + (1.0 if q is ln.weight else 0.0))
    return ffn, ln


@pytest.mark.parametrize("case", FFN_CASES, ids=[c[0] for c in FFN_CASES])
def test_fused_ffn_tail_vs_oracle(case):
    """y + FFN(LN(y)) fused == oracle (p = 0)"""
    from veloxseg_amd import functional as VF
    _, B, C, r, sp = case
    cm = VF.cpp_module()
    cm.set_fuse_blocks(True)
    ffn, ln = _ffn_modules(C, r, 0.0, 4)
    ffn, ln = ffn.cuda().train(), ln.cuda().train()
    y = torch.randn(B, C, *sp, generator=torch.Generator().manual_seed(8))
    yg = y.cuda().requires_grad_(True)
    out = VF.ffn_tail(yg, ln, ffn, 0.0)
    gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(9))
    out.backward(gy.cuda())
    torch.cuda.synchronize()
    sd = {"f." + k: v.detach().clone().cpu().requires_grad_(True) for k, v in ffn.state_dict().items()}
    lw, lb = ln.weight.detach().clone().cpu().requires_grad_(True), ln.bias.detach().clone().cpu().requires_grad_(True)
    yc = y.clone().requires_grad_(True)
    ref = yc + O.ffn(O.layernorm_cf(yc, lw, lb), sd, "f.", 0.0, True)
    ref.backward(gy)
    _close(out, ref, 2e-4, 2e-4, "ffn out")
    _close(yg.grad, yc.grad, 1e-3 * float(yc.grad.abs().max()), 1e-3, "ffn dy")
    for k, p in ffn.named_parameters():
        g = sd["f." + k].grad
        _close(p.grad, g, 2e-3 * float(g.abs().max()) + 1e-6, 2e-3, f"ffn d{k}")
    _close(ln.weight.grad, lw.grad, 2e-3 * float(lw.grad.abs().max()), 2e-3, "dgamma")
    _close(ln.bias.grad, lb.grad, 2e-3 * float(lb.grad.abs().max()), 2e-3, "dbeta")


@pytest.mark.parametrize("case", FFN_CASES[:2], ids=[c[0] for c in FFN_CASES[:2]])
def test_fused_ffn_tail_equals_per_operator_kernels_with_dropout(case):
    from veloxseg_amd import functional as VF
    _, B, C, r, sp = case
    cm = VF.cpp_module()
    res = {}
    try:
        for fused in (True, False):
            cm.set_fuse_blocks(fused)
            ffn, ln = _ffn_modules(C, r, 0.1, 4)
            ffn, ln = ffn.cuda().train(), ln.cuda().train()
            VF.manual_seed(91, "cuda")
            y = torch.randn(B, C, *sp, generator=torch.Generator().manual_seed(8)).cuda().requires_grad_(True)
            out = VF.ffn_tail(y, ln, ffn, 0.1)
            gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(9)).cuda()
            out.backward(gy)
            torch.cuda.synchronize()
            res[fused] = (out.detach().clone(), y.grad.clone(), {k: p.grad.clone() for k, p in list(ffn.named_parameters()) + [("ln." + n, q) for n, q in ln.named_parameters()]})
    finally:
        cm.set_fuse_blocks(True)
    _close(res[True][0], res[False][0], 1e-4, 1e-4, "out")
    _close(res[True][1], res[False][1], 1e-3 * float(res[False][1].abs().max()), 1e-3, "dy")
    for k in res[True][2]:
        g = res[False][2][k]
        _close(res[True][2][k], g, 2e-3 * float(g.abs().max()) + 1e-6, 2e-3, f"d{k}")


TZ_CASES = [
    # name, B, C, groups, spatial  (group widths 4 / 8 / 16; W % 4 == 0; ragged tiles; one w-block per row; anisotropic)
    ("L1_w4_32cube", 2, 16, 4, (32, 32, 32)),
    ("L2_w8_16cube", 2, 32, 4, (16, 16, 16)),
    ("L3_w8_8cube", 2, 64, 8, (8, 8, 8)),
    ("L4_w16_4cube", 2, 128, 8, (4, 4, 4)),
    ("L1_96", 1, 16, 4, (24, 24, 24)),
    ("L2_96", 1, 32, 4, (12, 12, 12)),
    ("aniso_ragged", 1, 16, 4, (10, 9, 20)),
    ("hecktor_L1", 1, 16, 4, (32, 32, 16)),
]


def test_jlc_block_backward_uses_the_pieces_mode_of_its_forward():
    """the operand image of a JLC block is laid out for the pieces mode in force at its forward; if the process-wide switch changes before the backward (another
    engine's precision, an A/B knob), the input gradient and the deferred weight gradients must still read it with the forward's mode (JLCFusedState.tz_pieces,
    vx_jlc_tz_bwd_ns / vx_jlc_wgrad_tz_ns) -- before, the backward indexed the image with the new mode's stride."""
    import veloxseg_amd.functional as VF
    from veloxseg_amd import _hip as H
    from veloxseg_amd.model.components.conv_blocks import JLC
    if VF.cpp_module() is None:
        pytest.skip("C++ operator module not built")
    H.LIB.load()

    def grads(switch):
        torch.manual_seed(3)
        m = JLC(16, groups=4).cuda().train()
        x = torch.randn(2, 16, 16, 16, 16, device="cuda", requires_grad=True)
        H.call("vx_jlc_tz_set_min_voxels", 0)
        H.call("vx_jlc_tz_set_pieces", 22)
        y = m(x)
        if switch:
            H.call("vx_jlc_tz_set_pieces", 3)            # (image stride 3 pieces instead of 2)
        (y * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
        torch.cuda.synchronize()
        return [x.grad.clone()] + [p.grad.clone() for p in m.parameters() if p.grad is not None]
    try:
        a, b = grads(False), grads(True)
        for u, v in zip(a, b):
            assert torch.allclose(u, v, rtol=1e-5, atol=1e-6 * float(u.abs().max())), float((u - v).abs().max())
    finally:
        H.call("vx_jlc_tz_set_pieces", 22)
        H.call("vx_jlc_tz_set_min_voxels", 1024)


@pytest.mark.parametrize("pieces", [3, 22, 1])
@pytest.mark.parametrize("case", TZ_CASES, ids=[c[0] for c in TZ_CASES])
def test_jlc_toeplitz_mfma_convs_vs_fp64_and_valu_kernels(case, pieces):
    """csrc/jlc_mfma.hip (the three grouped convs of conv_blocks.py:51-58 and their input gradient as Toeplitz GEMMs on the bf16 matrix pipe) against an fp64 torch
    convolution and against the fp32 VALU kernels of csrc/jlc.hip through the C ABI.  pieces = 3 (six piece products = the fp32 product): error within 3x the VALU
    kernels' own (fp32 summation noise), per-tile statistics summing to the tensor's; pieces = 1 (bf16 opt-in operands): 2^-8-level relative error, fp32 sums."""
    import torch.nn.functional as TF
    from veloxseg_amd import _hip as H
    _, B, C, G, (D, Hh, W) = case
    H.LIB.load()
    H.call("vx_jlc_tz_set_min_voxels", 0)
    H.call("vx_jlc_tz_set_pieces", pieces)
    try:
        assert H.query("vx_jlc_tz_ok", C, G, D, Hh, W) == 1
        torch.manual_seed(5)
        Cg = C // G
        x = torch.randn(B, C, D, Hh, W, device="cuda")
        ws = [torch.randn(C, Cg, k, k, k, device="cuda") * (1.0 / (Cg * k ** 3) ** 0.5) for k in (1, 3, 5)]
        bs = [torch.randn(C, device="cuda") * 0.1 for _ in range(3)]
        st = torch.cuda.current_stream().cuda_stream
        y_old, y_new = torch.empty(3, *x.shape, device="cuda"), torch.full((3, *x.shape), float("nan"), device="cuda")
        nt_old, nt_new = H.query("vx_jlc_ntiles", B, C, G, D, Hh, W), H.query("vx_jlc_tz_ntiles", C, G, D, Hh, W)
        p_old = torch.empty(3, B * C, nt_old, 2, device="cuda", dtype=torch.float64)
        p_new = torch.full((3, B * C, nt_new, 2), float("nan"), device="cuda", dtype=torch.float64)
        img = torch.empty(H.query("vx_jlc_tz_img_floats", C, G), device="cuda")
        H.call("vx_jlc_conv_fwd", H.P(x), H.P(ws[0]), H.P(ws[1]), H.P(ws[2]), H.P(bs[0]), H.P(bs[1]), H.P(bs[2]), y_old[0].data_ptr(), y_old[1].data_ptr(), y_old[2].data_ptr(),
               p_old.data_ptr(), B, C, G, D, Hh, W, st)
        H.call("vx_jlc_tz_prep", H.P(ws[0]), H.P(ws[1]), H.P(ws[2]), H.P(img), C, G, st)
        H.call("vx_jlc_tz_fwd", H.P(x), H.P(img), H.P(bs[0]), H.P(bs[1]), H.P(bs[2]), y_new[0].data_ptr(), y_new[1].data_ptr(), y_new[2].data_ptr(), p_new.data_ptr(),
               B, C, G, D, Hh, W, st)
        g = torch.randn(3, *x.shape, device="cuda")
        d_o = torch.randn_like(x)
        dx_old, dx_new = torch.empty_like(x), torch.full_like(x, float("nan"))
        H.call("vx_jlc_conv_bwd", g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), H.P(ws[0]), H.P(ws[1]), H.P(ws[2]), H.P(d_o), H.P(dx_old), B, C, G, D, Hh, W, st)
        H.call("vx_jlc_tz_bwd", g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), H.P(img), H.P(ws[0]), H.P(d_o), H.P(dx_new), B, C, G, D, Hh, W, st)
        torch.cuda.synchronize()
        exact = pieces in (3, 22)        # 3 bf16 pieces (six products) or two scaled fp16 pieces (22 significant bits, three products): fp32-level error
        for i, k in enumerate((1, 3, 5)):
            ref = TF.conv3d(x.double(), ws[i].double(), bs[i].double(), padding=k // 2, groups=G)
            sc = float(ref.abs().max())
            e_old, e_new = float((y_old[i].double() - ref).abs().max()) / sc, float((y_new[i].double() - ref).abs().max()) / sc
            if exact:
                assert e_new <= max(3.0 * e_old, 2e-6), (k, e_old, e_new)
            else:
                assert e_new <= 2e-2, (k, e_new)
            # per-tile (sum, sumsq) partials fold to the statistics of what the kernel STORED
            s_new = p_new[i].sum(1).view(B, C, 2)
            yk = y_new[i].double()
            assert torch.allclose(s_new[..., 0], yk.sum((2, 3, 4)), rtol=1e-5, atol=1e-3 * sc)
            assert torch.allclose(s_new[..., 1], (yk * yk).sum((2, 3, 4)), rtol=1e-5, atol=1e-3 * sc * sc)
        ref = d_o.double()
        for i, k in enumerate((1, 3, 5)):
            ref = ref + TF.conv_transpose3d(g[i].double(), ws[i].double(), None, padding=k // 2, groups=G)
        sc = float(ref.abs().max())
        e_old, e_new = float((dx_old.double() - ref).abs().max()) / sc, float((dx_new.double() - ref).abs().max()) / sc
        if exact:
            assert e_new <= max(3.0 * e_old, 2e-6), (e_old, e_new)
        else:
            assert e_new <= 2e-2, e_new
    finally:
        H.call("vx_jlc_tz_set_pieces", 22)          # (the library default)
        H.call("vx_jlc_tz_set_min_voxels", 1024)


CL_CASES = [
    # name, B, C, groups, spatial, scale of the inputs   (group widths 8 / 16, <= 8 voxels per axis; ragged last tile; values far outside fp16's range)
    ("L3_w8_8cube", 2, 64, 8, (8, 8, 8), 1.0),
    ("L4_w16_4cube", 2, 128, 8, (4, 4, 4), 1.0),
    ("L3_96_6cube", 1, 64, 8, (6, 6, 6), 1.0),
    ("L4_96_3cube", 3, 128, 8, (3, 3, 3), 1.0),
    ("aniso", 1, 64, 8, (8, 4, 6), 1.0),
    ("ragged_w8", 1, 32, 4, (5, 7, 3), 1.0),
    ("large_values", 1, 64, 8, (8, 8, 8), 3.0e4),
    ("tiny_values", 1, 128, 8, (4, 4, 4), 2.0e-6),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CL_CASES, ids=[c[0] for c in CL_CASES])
def test_jlc_channels_last_f16_convs_vs_fp64_and_valu_kernels(case):
    """csrc/jlc_cl.hip (the three grouped convs of conv_blocks.py:51-58 and their input gradient at the coarse levels, channels-last implicit GEMMs on the f16 matrix
    pipe with two scaled fp16 pieces per operand) against an fp64 torch convolution and against the fp32 VALU kernels of csrc/jlc.hip through the C ABI: error within
    3x the VALU kernels' own (fp32 summation noise), the per-wave (sum, sum of squares) partials fold to the statistics of the stored tensor, every element written."""
    import torch.nn.functional as TF
    from veloxseg_amd import _hip as H
    _, B, C, G, (D, Hh, W), scale = case
    H.LIB.load()
    assert H.query("vx_jlc_cl_ok", C, G, D, Hh, W) == 1
    torch.manual_seed(5)
    Cg = C // G
    x = torch.randn(B, C, D, Hh, W, device="cuda") * scale
    ws = [torch.randn(C, Cg, k, k, k, device="cuda") * (1.0 / (Cg * k ** 3) ** 0.5) for k in (1, 3, 5)]
    bs = [torch.randn(C, device="cuda") * 0.1 * scale for _ in range(3)]
    st = torch.cuda.current_stream().cuda_stream
    y_old, y_new = torch.empty(3, *x.shape, device="cuda"), torch.full((3, *x.shape), float("nan"), device="cuda")
    nt_old, nt_new = H.query("vx_jlc_ntiles", B, C, G, D, Hh, W), H.query("vx_jlc_cl_ntiles", C, G, D, Hh, W)
    p_old = torch.empty(3, B * C, nt_old, 2, device="cuda", dtype=torch.float64)
    p_new = torch.full((3, B * C, nt_new, 2), float("nan"), device="cuda", dtype=torch.float64)
    img = torch.empty(H.query("vx_jlc_cl_img_floats", C, G), device="cuda")
    H.call("vx_jlc_conv_fwd", H.P(x), H.P(ws[0]), H.P(ws[1]), H.P(ws[2]), H.P(bs[0]), H.P(bs[1]), H.P(bs[2]), y_old[0].data_ptr(), y_old[1].data_ptr(), y_old[2].data_ptr(),
           p_old.data_ptr(), B, C, G, D, Hh, W, st)
    H.call("vx_jlc_cl_prep", H.P(ws[0]), H.P(ws[1]), H.P(ws[2]), H.P(img), C, G, st)
    H.call("vx_jlc_cl_fwd", H.P(x), H.P(img), H.P(bs[0]), H.P(bs[1]), H.P(bs[2]), y_new[0].data_ptr(), y_new[1].data_ptr(), y_new[2].data_ptr(), p_new.data_ptr(),
           B, C, G, D, Hh, W, st)
    g = torch.randn(3, *x.shape, device="cuda") * scale
    g[1] *= 7.0                                   # (the three gradient tensors have scales of their own)
    g[0] *= 0.01
    d_o = torch.randn_like(x) * scale
    dx_old, dx_new = torch.empty_like(x), torch.full_like(x, float("nan"))
    H.call("vx_jlc_conv_bwd", g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), H.P(ws[0]), H.P(ws[1]), H.P(ws[2]), H.P(d_o), H.P(dx_old), B, C, G, D, Hh, W, st)
    H.call("vx_jlc_cl_bwd", g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), H.P(img), H.P(d_o), H.P(dx_new), B, C, G, D, Hh, W, st)
    torch.cuda.synchronize()
    for i, k in enumerate((1, 3, 5)):
        ref = TF.conv3d(x.double(), ws[i].double(), bs[i].double(), padding=k // 2, groups=G)
        sc = float(ref.abs().max())
        e_old, e_new = float((y_old[i].double() - ref).abs().max()) / sc, float((y_new[i].double() - ref).abs().max()) / sc
        assert e_new <= max(3.0 * e_old, 2e-6), (k, e_old, e_new)
        s_new = p_new[i].sum(1).view(B, C, 2)
        yk = y_new[i].double()
        assert torch.allclose(s_new[..., 0], yk.sum((2, 3, 4)), rtol=1e-5, atol=1e-3 * sc)
        assert torch.allclose(s_new[..., 1], (yk * yk).sum((2, 3, 4)), rtol=1e-5, atol=1e-3 * sc * sc)
    ref = d_o.double()
    for i, k in enumerate((1, 3, 5)):
        ref = ref + TF.conv_transpose3d(g[i].double(), ws[i].double(), None, padding=k // 2, groups=G)
    sc = float(ref.abs().max())
    e_old, e_new = float((dx_old.double() - ref).abs().max()) / sc, float((dx_new.double() - ref).abs().max()) / sc
    assert e_new <= max(3.0 * e_old, 2e-6), (e_old, e_new)


@pytest.mark.parametrize("pieces", [3, 22, 1, -3], ids=["3", "22", "1", "3_one_role"])
@pytest.mark.parametrize("case", [c for c in TZ_CASES if c[4][1] % 4 == 0 and c[4][2] <= 32], ids=[c[0] for c in TZ_CASES if c[4][1] % 4 == 0 and c[4][2] <= 32])
def test_jlc_toeplitz_mfma_weight_gradients_vs_fp64_and_valu_kernels(case, pieces):
    """vx_jlc_wgrad_tz (the three grouped-conv weight gradients of conv_blocks.py:51-58 in one matrix-pipe launch, csrc/jlc_mfma.hip) against torch's fp64 weight
    gradient and the fp32 VALU kernels (csrc/conv_wgrad.hip); the entry ACCUMULATES into dw (float atomics), checked by a non-zero start value.  pieces = 3 is the
    default instance with producer / consumer waves (512 threads, deeper staging); "3_one_role" the same arithmetic on the one-role kernel (vx_jlc_wgrad_tz_set_spec(0))."""
    from veloxseg_amd import _hip as H
    _, B, C, G, (D, Hh, W) = case
    H.LIB.load()
    H.call("vx_jlc_wgrad_tz_set_spec", 0 if pieces < 0 else 1)
    pieces = abs(pieces)
    H.call("vx_jlc_tz_set_pieces", pieces)
    H.call("vx_jlc_wgrad_tz_set_f16", 1)          # pieces = 22: the two-fp16-piece weight-gradient instance (an A/B variant: the default under 22 is three bf16 pieces)
    try:
        assert H.query("vx_jlc_wgrad_tz_ok", C, G, D, Hh, W) == 1
        torch.manual_seed(11)
        Cg = C // G
        x = torch.randn(B, C, D, Hh, W, device="cuda")
        g = torch.randn(3, B, C, D, Hh, W, device="cuda")
        shapes = [(C, Cg, k, k, k) for k in (1, 3, 5)]
        init = [torch.randn(s_, device="cuda") for s_ in shapes]
        d_new = [t.clone() for t in init]
        d_old = [torch.zeros(s_, device="cuda") for s_ in shapes]
        st = torch.cuda.current_stream().cuda_stream
        H.call("vx_jlc_wgrad_tz", H.P(x), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), H.P(d_new[0]), H.P(d_new[1]), H.P(d_new[2]), B, C, G, D, Hh, W, st)
        H.call("vx_conv3d_bwd_weight_tiled", H.P(x), None, 0, g[1].data_ptr(), H.P(d_old[1]), None, B, C, D, Hh, W, C, 3, 1, 1, G, 1, st)
        H.call("vx_conv3d_bwd_weight_tiled", H.P(x), None, 0, g[2].data_ptr(), H.P(d_old[2]), None, B, C, D, Hh, W, C, 5, 1, 2, G, 1, st)
        torch.cuda.synchronize()
        for i, k in enumerate((1, 3, 5)):
            ref = torch.nn.grad.conv3d_weight(x.double(), shapes[i], g[i].double(), padding=k // 2, groups=G)
            sc = float(ref.abs().max())
            e_new = float(((d_new[i] - init[i]).double() - ref).abs().max()) / sc
            if pieces in (3, 22):
                e_old = float((d_old[i].double() - ref).abs().max()) / sc if i else 0.0
                assert e_new <= max(3.0 * e_old, 3e-6), (k, e_old, e_new)
            else:
                assert e_new <= 2e-2, (k, e_new)
    finally:
        H.call("vx_jlc_tz_set_pieces", 22)          # (the library default)
        H.call("vx_jlc_wgrad_tz_set_f16", 0)
        H.call("vx_jlc_wgrad_tz_set_spec", 1)

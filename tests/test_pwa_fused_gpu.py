"""GPU parity of the fused PWA chains (csrc/pwa_fused.hip): LN + q / k / v of every modality in one launch ("pre"), mix conv + residual + LN + FFN in
one launch ("post", at the levels csrc/mlp.hip does not cover), PatchMerging (gather + LN(8C) + reduction) in one launch, and the grouped
weight-gradient launch behind them (pointwise.hip vx_pw_wgrad_group).  One transformer layer (reference PWA.py:444-511: block + PatchMerging),
forward and every gradient,
  (a) against the CPU oracle with dropout off (fp32: outputs 2e-4, gradients 2e-3 of the gradient's scale), and
  (b) against the per-operator kernels of the same library with dropout ON (same Philox masks => equal to summation-order noise)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import veloxseg_oracle as O  # noqa: E402  (checker only)

# name, B, M, C, grid, window, heads, dim_head, ffn ratio, downsample
CASES = [
    ("L1_16cube_T4", 2, 2, 16, (16, 16, 16), (2, 2, 2), 1, 4, 3, True),
    ("L2_16cube_T4", 2, 2, 32, (16, 16, 16), (8, 8, 8), 2, 8, 3, True),
    ("L3_8cube", 2, 2, 64, (8, 8, 8), (4, 4, 4), 2, 8, 2, True),
    ("L4_4cube", 3, 2, 128, (4, 4, 4), (4, 4, 4), 4, 16, 2, False),
    ("L3_96_6cube", 2, 2, 64, (6, 6, 6), (3, 3, 3), 2, 8, 2, True),
    ("L4_96_3cube", 2, 2, 128, (3, 3, 3), (3, 3, 3), 4, 16, 2, False),
    ("L2_96_12cube", 1, 2, 32, (12, 12, 12), (6, 6, 6), 2, 8, 3, True),
    ("brats_M1_L3", 2, 1, 64, (8, 8, 8), (4, 4, 4), 2, 8, 2, True),
    ("aniso_L3", 2, 2, 64, (8, 8, 4), (4, 4, 2), 2, 8, 2, True),
    ("L2_T4_partial_tile", 1, 2, 32, (12, 14, 14), (6, 7, 7), 2, 8, 3, False),      # 2352 voxels = 36.75 tiles of 64
]


def _close(a, b, atol, rtol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    if not bool((err <= tol).all()):
        i = int((err - tol).argmax())
        raise AssertionError(f"{what}: max abs err {float(err.max()):.3e} (ref max {float(b.abs().max()):.3e}); worst idx {i}: got {float(a.flatten()[i]):.6e} "
                             f"want {float(b.flatten()[i]):.6e}; bad frac {float((err > tol).double().mean()):.3e}")


def _layer(case, p, seed=3):
    from veloxseg_amd import functional as VF
    from veloxseg_amd.model.components.PWA import Transformer_BasicLayer
    _, B, M, C, grid, win, heads, dh, ratio, down = case
    torch.manual_seed(seed)
    VF.reset_dropout_sites()
    layer = Transformer_BasicLayer(input_size=list(grid), in_channels=[C] * M, depth=1, min_big_window_size=list(win), min_small_window_size=[1, 1, 1], num_heads=heads,
                                   min_dim_head=dh, attn_drop=p, proj_drop=p, ffn_expansion_ratio=ratio, do_downsample=down)
    with torch.no_grad():
        for n, q in layer.named_parameters():
            if q.dim() > 1:
                q.copy_(torch.randn_like(q) * (0.5 / max(1.0, float(q[0].numel())) ** 0.5 if "table" not in n else 0.3))
            elif n.endswith("weight"):
                q.copy_(1.0 + 0.2 * torch.randn_like(q))
            else:
                q.copy_(0.1 * torch.randn_like(q))
    return layer


def _run(layer, xs_cpu, gy_cpu, fused):
    from veloxseg_amd import functional as VF
    old = VF.USE_PWA_FUSED
    VF.USE_PWA_FUSED = fused
    try:
        layer.zero_grad(set_to_none=True)
        xs = [x.cuda().requires_grad_(True) for x in xs_cpu]
        VF.manual_seed(77, "cuda")
        VF.advance_rng(torch.device("cuda"))
        outs, down = layer(xs)
        ys = list(outs) + (list(down) if down is not None else [])
        torch.autograd.backward(ys, [g.cuda() for g in gy_cpu[:len(ys)]])
        torch.cuda.synchronize()
        return [y.detach().cpu() for y in ys], [x.grad.detach().cpu() for x in xs], {n: q.grad.detach().cpu().clone() for n, q in layer.named_parameters() if q.grad is not None}
    finally:
        VF.USE_PWA_FUSED = old


def _inputs(case):
    _, B, M, C, grid, *_ = case
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(B, C, *grid, generator=g) for _ in range(M)]
    gy = [torch.randn(B, C, *grid, generator=g) for _ in range(M)] + [torch.randn(B, 2 * C, *[v // 2 for v in grid], generator=g) for _ in range(M)]
    return xs, gy


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_fused_layer_vs_oracle(case):
    name, B, M, C, grid, win, heads, dh, ratio, down = case
    layer = _layer(case, 0.0).cuda().train()
    xs, gy = _inputs(case)
    outs, dxs, grads = _run(layer, xs, gy, True)
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in layer.state_dict().items()}
    plan = O.plan_pwa(list(grid), list(win), [1, 1, 1], 2, heads, dh, C)
    xc = [x.clone().requires_grad_(True) for x in xs]
    ref = O.pwa_block(xc, sd, "blocks.0.", plan, {"attn": 0.0, "proj": 0.0}, True)
    if down:
        ref = ref + [O.patch_merging(ref[m], sd, f"downs.{m}.") for m in range(M)]
    torch.autograd.backward(ref, gy[:len(ref)])
    for i, (a, b) in enumerate(zip(outs, ref)):
        _close(a, b, 2e-4, 2e-4, f"{name} output {i}")
    for m in range(M):
        _close(dxs[m], xc[m].grad, 2e-3 * float(xc[m].grad.abs().max()), 2e-3, f"{name} dx[{m}]")
    for k, g in grads.items():
        r = sd[k].grad
        assert r is not None, k
        _close(g, r, 2e-3 * float(r.abs().max()) + 5e-6, 2e-3, f"{name} d{k}")      # (+5e-6: the key bias of a single-modality block has a zero gradient, both sides hold round-off)


@pytest.mark.parametrize("case", [CASES[i] for i in (0, 1, 2, 3, 5, 7, 9)], ids=[CASES[i][0] for i in (0, 1, 2, 3, 5, 7, 9)])
def test_fused_layer_equals_per_operator_kernels_with_dropout(case):
    """dropout 0.1 on every site: the fused launches regenerate the masks of the per-operator kernels, so the two paths agree to summation-order noise"""
    name = case[0]
    layer = _layer(case, 0.1).cuda().train()
    xs, gy = _inputs(case)
    a_out, a_dx, a_g = _run(layer, xs, gy, True)
    b_out, b_dx, b_g = _run(layer, xs, gy, False)
    for i, (a, b) in enumerate(zip(a_out, b_out)):
        _close(a, b, 1e-4, 1e-4, f"{name} output {i}")
        assert float((b == 0).float().mean()) < 0.5
    for m in range(len(a_dx)):
        _close(a_dx[m], b_dx[m], 5e-4 * float(b_dx[m].abs().max()), 5e-4, f"{name} dx[{m}]")
    assert set(a_g) == set(b_g)
    for k in a_g:
        _close(a_g[k], b_g[k], 1e-3 * float(b_g[k].abs().max()) + 5e-6, 1e-3, f"{name} d{k}")


def test_fused_paths_are_taken():
    """the shapes of the headline network reach the fused kernels (a silent fall-back to the per-operator launches would keep every other test green)"""
    from veloxseg_amd import functional as VF
    from veloxseg_amd import _hip as H
    calls = []
    real = H.LIB.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)
    H.LIB.call = spy
    try:
        for case in (CASES[0], CASES[2]):
            layer = _layer(case, 0.0).cuda().train()
            xs, gy = _inputs(case)
            calls.clear()
            _run(layer, xs, gy, True)
            assert calls.count("vx_ln_pw_fwd") == 2 and calls.count("vx_ln_pw_bwd") == 2, (case[0], calls)        # the block's pre + PatchMerging
            assert "vx_pw_wgrad_group" in calls
            if case[0] == "L3_8cube":
                assert calls.count("vx_pwa_post_fwd") == 1 and calls.count("vx_pwa_post_bwd") == 1, calls
    finally:
        H.LIB.call = real

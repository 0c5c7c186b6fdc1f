"""CPU ORACLE for the VeloxSeg forward/backward path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product path (veloxseg_amd/) never imports it and has no CPU fallback.

What it is: a functional (no nn.Module) fp32 PyTorch-CPU restatement of the reference algorithm,
driven by a reference-format state_dict.  Each function cites the reference file:line it follows
(paths relative to /root/reference).  It is a floating-point path, so the oracle is a torch fp32
reference (tier rule 3); autograd differentiates it, which gives the backward oracle for free.

Parity pinning: the reference's own tests pin only utils/runtime.py helpers
(tests/test_runtime_helpers.py:63-75,87-111).  Everything else is pinned here against outputs of
the reference itself, imported in the build container with a MONAI stand-in
(tests/golden/make_golden.py -> tests/golden/*.pt, checked by tests/test_oracle_golden.py).
MONAI 1.5.0 (requirements.txt:3) is absent from /root/reference; the semantics assumed for
PatchEmbed / DiceLoss / DropPath / trunc_normal_ / get_act_layer are restated in make_golden.py and
SURVEY.md Appendix A6.

Deliberate differences of FORM from the reference (same maths, proven by the goldens):
  * window gather = max-pool of the whole volume, then partition (reference pools per window,
    model/components/PWA.py:106-140);
  * window scatter and deep-supervision up-sampling use explicit separable 1-D align_corners
    interpolation matrices instead of F.interpolate (PWA.py:177-200, VeloxSeg.py:177-184);
  * LayerNorm is evaluated once per block input, not once per q/k/v (PWA.py:341-343).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# host-side geometry (model/components/PWA.py:56-86)
# --------------------------------------------------------------------------------------------
def plan_pwa(grid: Sequence[int], big: Sequence[int], small: Sequence[int], scale_factor: int,
             heads: int, min_dim_head: int, channels: int) -> dict:
    """Scales and channel split of one PWA layer.  `while (bw <= input).any()` (PWA.py:67)."""
    bws, sws = [], []
    bw, sw = list(big), list(small)
    while any(b <= g for b, g in zip(bw, grid)):
        bws.append(list(bw))
        sws.append(list(sw))
        bw = [b * scale_factor for b in bw]
        sw = [s * scale_factor for s in sw]
    need = len(bws) * heads * min_dim_head
    ch_qk = need
    ch_v = math.ceil(channels / need) * need                       # PWA.py:74-76
    n = [big[i] // small[i] for i in range(3)]                     # PWA.py:42
    for b in bws:
        if any(g // x == 0 or g % x for g, x in zip(grid, b)):
            raise ValueError(f"PWA window {b} does not tile grid {list(grid)}")
    return dict(grid=list(grid), big=bws, small=sws, n=n, heads=heads, nb=len(bws), ch_qk=ch_qk, ch_v=ch_v,
                c_qk=ch_qk // (len(bws) * heads), c_v=ch_v // (len(bws) * heads),
                nwin=[[g // b for g, b in zip(grid, bb)] for bb in bws])


# --------------------------------------------------------------------------------------------
# norms / activations (attention_utils.py:29-43; common_function.py:63-66,93-94)
# --------------------------------------------------------------------------------------------
def layernorm_cf(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    xn = (x - u) / torch.sqrt(s + eps)
    shape = (1, -1) + (1,) * (x.ndim - 2)
    return w.view(shape) * xn + b.view(shape)


def instnorm(x: Tensor, eps: float = 1e-5) -> Tensor:
    """InstanceNorm3d, affine=False, no running stats, biased variance (torch default)."""
    dims = tuple(range(2, x.ndim))
    mu = x.mean(dims, keepdim=True)
    var = (x - mu).pow(2).mean(dims, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps)


def gelu(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * 0.7071067811865476))


def dropout(x: Tensor, p: float, training: bool) -> Tensor:
    return F.dropout(x, p, training) if (training and p > 0) else x


# --------------------------------------------------------------------------------------------
# interpolation matrices (align_corners=True), mirrors aten's fp32 index/lambda arithmetic
# --------------------------------------------------------------------------------------------
def interp_matrix(n_in: int, n_out: int) -> Tensor:
    A = torch.zeros(n_out, n_in, dtype=torch.float32)
    if n_out == 1:
        A[0, 0] = 1.0
        return A
    scale = torch.tensor((n_in - 1) / (n_out - 1), dtype=torch.float32) if n_out > 1 else torch.tensor(0.0)
    for j in range(n_out):
        src = scale * j
        i0 = int(src)
        lam = float(src - i0)
        i1 = i0 + (1 if i0 < n_in - 1 else 0)
        A[j, i0] += 1.0 - lam
        A[j, i1] += lam
    return A


def upsample_trilinear(x: Tensor, size: Sequence[int]) -> Tensor:
    """F.interpolate(x, size, mode='trilinear', align_corners=True) (VeloxSeg.py:177-184)."""
    if list(x.shape[2:]) == list(size):
        return x
    Ad, Ah, Aw = (interp_matrix(x.shape[2 + i], size[i]).to(x.device) for i in range(3))
    x = torch.einsum("bcdhw,Dd->bcDhw", x, Ad)
    x = torch.einsum("bcdhw,Hh->bcdHw", x, Ah)
    x = torch.einsum("bcdhw,Ww->bcdhW", x, Aw)
    return x


# --------------------------------------------------------------------------------------------
# PWA (model/components/PWA.py)
# --------------------------------------------------------------------------------------------
def gather_windows(x: Tensor, plan: dict, c: int) -> Tensor:
    """(B, nb*h*c, g0,g1,g2) -> (B, h, sumN, l, c).  PWA.py:106-140; channel order (bswin, head, c) :111."""
    B = x.shape[0]
    h, nb, n = plan["heads"], plan["nb"], plan["n"]
    x = x.view(B, nb, h * c, *x.shape[2:])
    outs = []
    for i in range(nb):
        s = plan["small"][i]
        p = F.max_pool3d(x[:, i], kernel_size=s, stride=s) if any(v > 1 for v in s) else x[:, i]
        N0, N1, N2 = plan["nwin"][i]
        p = p.view(B, h, c, N0, n[0], N1, n[1], N2, n[2])
        p = p.permute(0, 1, 3, 5, 7, 4, 6, 8, 2).reshape(B, h, N0 * N1 * N2, n[0] * n[1] * n[2], c)
        outs.append(p)
    return torch.cat(outs, 2)


def scatter_windows(tok: Tensor, plan: dict, c: int) -> Tensor:
    """(B, h, sumN, l, c) -> (B, nb*h*c, g).  Per-window trilinear up-sampling, align_corners=True (PWA.py:177-200)."""
    B = tok.shape[0]
    h, nb, n = plan["heads"], plan["nb"], plan["n"]
    g = plan["grid"]
    outs, off = [], 0
    for i in range(nb):
        N0, N1, N2 = plan["nwin"][i]
        N = N0 * N1 * N2
        s = plan["small"][i]
        t = tok[:, :, off:off + N].reshape(B, h, N0, N1, N2, n[0], n[1], n[2], c)
        off += N
        A0, A1, A2 = (interp_matrix(n[k], n[k] * s[k]).to(tok.device) for k in range(3))
        t = torch.einsum("bhxyzijkc,Ii->bhxyzIjkc", t, A0)
        t = torch.einsum("bhxyzijkc,Jj->bhxyziJkc", t, A1)
        t = torch.einsum("bhxyzijkc,Kk->bhxyzijKc", t, A2)
        t = t.permute(0, 1, 8, 2, 5, 3, 6, 4, 7).reshape(B, h * c, g[0], g[1], g[2])
        outs.append(t)
    return torch.cat(outs, 1)


def relative_position_index(n: Sequence[int]) -> Tensor:
    """attention_utils.py:83-101 (int64 buffer in the state dict)."""
    coords = torch.stack(torch.meshgrid(*[torch.arange(k) for k in n], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += n[0] - 1
    rel[:, :, 1] += n[1] - 1
    rel[:, :, 2] += n[2] - 1
    rel[:, :, 0] *= (2 * n[1] - 1) * (2 * n[2] - 1)
    rel[:, :, 1] *= 2 * n[2] - 1
    return rel.sum(-1)


def relative_bias(table: Tensor, index: Tensor, l: int) -> Tensor:
    """(heads, l, l).  attention_utils.py:120-125."""
    return table[index[:l, :l].reshape(-1)].view(l, l, -1).permute(2, 0, 1)


def window_attention(q: Tensor, k: Tensor, v: Tensor, bias: Tensor, M: int, p_drop: float = 0.0,
                     training: bool = False) -> Tensor:
    """q,k: (B,h,N,M*l,cqk), v: (B,h,N,M*l,cv); same bias on every (m,m') block (PWA.py:308-327)."""
    c = q.shape[-1]
    s = torch.einsum("bhnic,bhnjc->bhnij", q, k) / (c ** 0.5)
    s = s + bias.repeat(1, M, M)[None, :, None]
    w = torch.softmax(s, -1)
    w = dropout(w, p_drop, training)
    return torch.einsum("bhnij,bhnjc->bhnic", w, v)


def pwa_attention_module(xs: List[Tensor], sd: Dict[str, Tensor], pre: str, plan: dict, drops: dict, training: bool) -> List[Tensor]:
    """MultiModal_Paired_Windows_Attention.forward (PWA.py:329-379): returns x_m + Drop(mix(scatter(attn)))."""
    M = len(xs)
    qs, ks, vs = [], [], []
    for m in range(M):
        xn = layernorm_cf(xs[m], sd[f"{pre}input_norms.{m}.weight"], sd[f"{pre}input_norms.{m}.bias"])
        def proj(j):
            w = sd[f"{pre}qkv_proj.{m}.{j}.weight"]
            b = sd.get(f"{pre}qkv_proj.{m}.{j}.bias")
            return F.conv3d(xn, w, b)
        qs.append(gather_windows(proj(0), plan, plan["c_qk"]))
        ks.append(gather_windows(proj(1), plan, plan["c_qk"]))
        vs.append(gather_windows(proj(2), plan, plan["c_v"]))
    q, k, v = torch.cat(qs, 3), torch.cat(ks, 3), torch.cat(vs, 3)
    l = qs[0].shape[3]
    bias = relative_bias(sd[f"{pre}position_embedding.relative_position_bias_table"],
                         sd[f"{pre}position_embedding.relative_position_index"], l)
    a = window_attention(q, k, v, bias, M, drops["attn"], training)
    outs = []
    for m in range(M):
        sc = scatter_windows(a[:, :, :, m * l:(m + 1) * l], plan, plan["c_v"])
        mix = F.conv3d(sc, sd[f"{pre}mix_channels.{m}.weight"], sd[f"{pre}mix_channels.{m}.bias"])
        outs.append(xs[m] + dropout(mix, drops["proj"], training))
    return outs


def ffn(x: Tensor, sd, pre: str, p: float, training: bool) -> Tensor:
    """attention_utils.py:45-71."""
    h = gelu(F.conv3d(x, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"]))
    h = dropout(h, p, training)
    y = F.conv3d(h, sd[pre + "linear2.weight"], sd[pre + "linear2.bias"])
    return dropout(y, p, training)


def pwa_block(xs: List[Tensor], sd, pre: str, plan: dict, drops: dict, training: bool) -> List[Tensor]:
    """Paired_Windows_TransformerBlock.forward (PWA.py:433-439): y = x + attn(x) [= 2x + mix]; z = y + FFN(LN(y))."""
    attns = pwa_attention_module(xs, sd, pre + "attn.", plan, drops, training)
    out = []
    for m in range(len(xs)):
        y = xs[m] + attns[m]
        yn = layernorm_cf(y, sd[f"{pre}norms.{m}.weight"], sd[f"{pre}norms.{m}.bias"])
        out.append(y + ffn(yn, sd, f"{pre}ffns.{m}.", drops["proj"], training))
    return out


def space_to_depth2(x: Tensor) -> Tensor:
    """PatchMerging.faeture_sample (attention_utils.py:144-159): sub-volume index outermost in channels."""
    parts = [x[:, :, i::2, j::2, k::2] for i in (0, 1) for j in (0, 1) for k in (0, 1)]
    return torch.cat(parts, 1)


def patch_merging(x: Tensor, sd, pre: str) -> Tensor:
    """attention_utils.py:161-168."""
    y = layernorm_cf(space_to_depth2(x), sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    return F.conv3d(y, sd[pre + "reduction.weight"])


# --------------------------------------------------------------------------------------------
# conv blocks (model/components/conv_blocks.py)
# --------------------------------------------------------------------------------------------
def down_conv(x: Tensor, sd, pre: str, p: int) -> Tensor:
    """DownConv (conv_blocks.py:4-21): k=2p-1, stride p, pad p-1, then IN."""
    return instnorm(F.conv3d(x, sd[pre + "down.weight"], sd[pre + "down.bias"], stride=p, padding=p - 1))


def up_conv(x: Tensor, sd, pre: str) -> Tensor:
    """UpConv (conv_blocks.py:23-39): ConvTranspose3d k2 s2 then IN."""
    return instnorm(F.conv_transpose3d(x, sd[pre + "up.weight"], sd[pre + "up.bias"], stride=2))


def jlc(x: Tensor, sd, pre: str, groups: int, p_drop: float, training: bool) -> Tensor:
    """JLC.forward (conv_blocks.py:72-75)."""
    acc = x
    for i, k in enumerate((1, 3, 5)):
        w, b = sd[f"{pre}spatial_convs.{i}.0.weight"], sd[f"{pre}spatial_convs.{i}.0.bias"]
        assert w.shape[-1] == k
        acc = acc + gelu(instnorm(F.conv3d(x, w, b, padding=k // 2, groups=groups)))
    h = gelu(F.conv3d(instnorm(acc), sd[pre + "channel_conv.1.weight"], sd[pre + "channel_conv.1.bias"]))
    y = F.conv3d(h, sd[pre + "channel_conv.3.weight"], sd[pre + "channel_conv.3.bias"])
    return acc + dropout(y, p_drop, training)


def jlc_layer(x, sd, pre, depth, groups, p_drop, training):
    for d in range(depth):
        x = jlc(x, sd, f"{pre}{d}.", groups, p_drop, training)
    return x


def pixel_shuffle3d(x: Tensor, s: int) -> Tensor:
    """superpixel.py:16: 'b (c s1 s2 s3) d h w -> b c (d s1) (h s2) (w s3)'."""
    B, C, D, H, W = x.shape
    c = C // (s ** 3)
    x = x.view(B, c, s, s, s, D, H, W).permute(0, 1, 5, 2, 6, 3, 7, 4)
    return x.reshape(B, c, D * s, H * s, W * s)


def gram(x: Tensor) -> Tensor:
    """get_pram_matrix (common_function.py:8-14): G / (C*H*W*D)."""
    B, C = x.shape[:2]
    f = x.reshape(B, C, -1)
    return torch.bmm(f, f.transpose(1, 2)) / (C * f.shape[2])


# --------------------------------------------------------------------------------------------
# the network
# --------------------------------------------------------------------------------------------
class OracleConfig:
    """Accepts exactly the reference constructor kwargs (model/VeloxSeg.py:64-94)."""

    def __init__(self, input_size, patch_size, in_ch, n_classes=2, base_ch=16, conv_depths=(1, 1, 1, 1),
                 kernel_sizes=(1, 3, 5), min_dim_group=(4, 8, 8, 16), conv_expansion_factor=(3, 3, 2, 2),
                 attn_base_ch=16, depths=(2, 2, 2, 2), min_big_window_sizes=((3, 3, 3), (6, 6, 6), (3, 3, 3), (3, 3, 3)),
                 min_small_window_sizes=((1, 1, 1),) * 4, min_dim_head=(4, 8, 8, 16), scale_factors=(2, 2, 2, 2),
                 num_heads=(1, 2, 2, 4), attn_drop=0.1, proj_drop=0.1, drop_path=0, ffn_expansion_ratio=(3, 3, 2, 2),
                 act_layer="GELU", norm_layer=None, patch_norm=False, qkv_bias=True, conv_drop=0.0,
                 deep_supervision=True, spatial_dim=3):
        assert spatial_dim == 3 and tuple(kernel_sizes) == (1, 3, 5) and drop_path == 0 and not patch_norm
        self.input_size = list(input_size)
        self.patch_size = patch_size
        self.in_ch = list(in_ch)
        self.M = len(in_ch)
        self.n_classes = n_classes
        self.base_ch = base_ch
        self.attn_base_ch = attn_base_ch
        self.conv_depths = list(conv_depths)
        self.depths = list(depths)
        self.groups = [base_ch * 2 ** i // min_dim_group[i] for i in range(4)]
        self.drops = dict(attn=attn_drop, proj=proj_drop, conv=conv_drop)
        self.deep_supervision = deep_supervision
        grid = [s // patch_size for s in input_size]
        self.grids, self.plans = [], []
        for L in range(4):
            self.grids.append(list(grid))
            self.plans.append(plan_pwa(grid, min_big_window_sizes[L], min_small_window_sizes[L], scale_factors[L],
                                       num_heads[L], min_dim_head[L], attn_base_ch * 2 ** L))
            grid = [g // 2 for g in grid]


def encoder(x: Tensor, sd, cfg: OracleConfig, training: bool):
    """Encoder.forward (model/Encoder.py:339-367) incl. Transformer_Encoder.forward (:190-204)."""
    M = cfg.M
    xs = list(torch.chunk(x, M, dim=1))                              # Encoder.py:192 (equal chunks)
    p = cfg.patch_size
    cur = []
    for m in range(M):
        w, b = sd[f"encoder.encoder_attn.patch_embeds.{m}.proj.weight"], sd[f"encoder.encoder_attn.patch_embeds.{m}.proj.bias"]
        cur.append(dropout(F.conv3d(xs[m], w, b, stride=p), cfg.drops["proj"], training))
    attn_feats = []
    for L in range(4):
        for d in range(cfg.depths[L]):
            cur = pwa_block(cur, sd, f"encoder.encoder_attn.layers.{L}.blocks.{d}.", cfg.plans[L], cfg.drops, training)
        attn_feats.append(cur)
        if L < 3:
            cur = [patch_merging(cur[m], sd, f"encoder.encoder_attn.layers.{L}.downs.{m}.") for m in range(M)]
    encs = []
    prev = x
    for L in range(4):
        a = instnorm(F.conv3d(torch.cat(attn_feats[L], 1), sd[f"encoder.attn2conv_{L + 1}.0.weight"],
                              sd[f"encoder.attn2conv_{L + 1}.0.bias"]))
        y = down_conv(prev, sd, f"encoder.encoder_conv.down{L + 1}.", p if L == 0 else 2) + a
        prev = jlc_layer(y, sd, f"encoder.encoder_conv.layer{L + 1}.", cfg.conv_depths[L], cfg.groups[L], cfg.drops["conv"], training)
        encs.append(prev)
    return attn_feats, encs


def decoder_trunk(e: List[Tensor], sd, pre: str, cfg: OracleConfig, training: bool):
    """Shared trunk of Seg_Decoder / RC_Decoder (model/Decoder.py:160-164, 85-88)."""
    up3 = jlc_layer(e[2] + up_conv(e[3], sd, pre + "layer_up3."), sd, pre + "layer3.", cfg.conv_depths[2], cfg.groups[2], cfg.drops["conv"], training)
    up2 = jlc_layer(e[1] + up_conv(up3, sd, pre + "layer_up2."), sd, pre + "layer2.", cfg.conv_depths[1], cfg.groups[1], cfg.drops["conv"], training)
    up1 = jlc_layer(e[0] + up_conv(up2, sd, pre + "layer_up1."), sd, pre + "layer1.", cfg.conv_depths[0], cfg.groups[0], cfg.drops["conv"], training)
    return up1, up2, up3


def forward(x: Tensor, sd: Dict[str, Tensor], cfg: OracleConfig, training: bool):
    """VeloxSeg.forward (model/VeloxSeg.py:186-226).  Train -> list; eval -> logits."""
    attn_feats, encs = encoder(x, sd, cfg, training)
    up1, up2, up3 = decoder_trunk(encs, sd, "decoder.", cfg, training)
    s = cfg.patch_size
    main = pixel_shuffle3d(F.conv3d(up1, sd["decoder.out_conv1.0.weight"], sd["decoder.out_conv1.0.bias"], padding=1), s)
    if not training:
        return main
    preds = [main]
    if cfg.deep_supervision:
        preds.append(F.conv3d(up2, sd["decoder.out_conv2.weight"], sd["decoder.out_conv2.bias"]))
        preds.append(F.conv3d(up3, sd["decoder.out_conv3.weight"], sd["decoder.out_conv3.bias"]))
        preds.append(F.conv3d(encs[3], sd["decoder.out_conv4.weight"], sd["decoder.out_conv4.bias"]))
    preds = [upsample_trilinear(t, cfg.input_size) for t in preds]
    rcs, grams = [], []
    for m in range(cfg.M):
        pre = f"rc_decoders.{m}."
        e = []
        for L in range(4):
            cat = torch.cat([attn_feats[L][m], encs[L]], 1)
            e.append(instnorm(F.conv3d(cat, sd[f"{pre}enc2rc_{L + 1}.0.weight"], sd[f"{pre}enc2rc_{L + 1}.0.bias"])))
        r1, _, _ = decoder_trunk(e, sd, pre, cfg, training)
        rcs.append(pixel_shuffle3d(F.conv3d(r1, sd[pre + "out_conv.0.weight"], sd[pre + "out_conv.0.bias"], padding=1), s))
        grams.append(gram(r1))
    return preds + [torch.cat(rcs, 1)] + [gram(up1)] + grams


# --------------------------------------------------------------------------------------------
# loss (utils/loss.py:30-66; utils/runtime.py:125-174)
# --------------------------------------------------------------------------------------------
def normalized_deep_loss_weights(configured, output_count):
    if output_count <= 0:
        raise ValueError("output_count must be greater than 0")
    w = [float(v) for v in configured]
    if not w:
        raise ValueError("deep_Loss_weight must contain at least one value")
    if sum(w) == 0:
        raise ValueError("deep_Loss_weight sum must be non-zero")
    if len(w) != output_count:
        if all(v == w[0] for v in w):
            return [1.0 / output_count] * output_count
        raise ValueError("deep_Loss_weight length must match model deep-supervision outputs unless all configured weights are equal")
    t = sum(w)
    return [v / t for v in w]


def veloxseg_output_layout(output_count, num_modal):
    tail = 2 + int(num_modal)
    if output_count <= tail:
        raise ValueError(f"VeloxSeg output count {output_count} is too small for {num_modal} modality reconstruction outputs")
    n = output_count - tail
    return {"seg": (0, n), "reconstruction": n, "decoder_gram": n + 1, "teacher_grams": tuple(range(n + 2, n + 2 + int(num_modal)))}


def dice_loss(logits: Tensor, labels: Tensor) -> Tensor:
    """MONAI DiceLoss(include_background=False, to_onehot_y=True, softmax=True) (utils/loss.py:18-20; A6)."""
    C = logits.shape[1]
    p = torch.softmax(logits, 1)
    t = F.one_hot(labels.squeeze(1).long(), C).movedim(-1, 1).to(p.dtype)
    p, t = p[:, 1:], t[:, 1:]
    dims = (2, 3, 4)
    inter = (p * t).sum(dims)
    den = p.sum(dims) + t.sum(dims)
    return (1.0 - (2.0 * inter + 1e-5) / (den + 1e-5)).mean()


def seg_loss(logits: Tensor, labels: Tensor) -> Tensor:
    return F.cross_entropy(logits, labels.squeeze(1).long()) + dice_loss(logits, labels)


def loss(outputs: List[Tensor], labels: Tensor, sr_labels: Tensor, num_modal: int, loss_cfg: dict) -> Tensor:
    lay = veloxseg_output_layout(len(outputs), num_modal)
    a, b = lay["seg"]
    w = normalized_deep_loss_weights(loss_cfg["deep_Loss_weight"], b - a)
    total = outputs[0].new_zeros(())
    for wi, o in zip(w, outputs[a:b]):
        total = total + wi * seg_loss(o, labels)
    rc = F.mse_loss(outputs[lay["reconstruction"]], sr_labels)
    feat = 0
    for ti in lay["teacher_grams"]:
        feat = feat + F.mse_loss(outputs[lay["decoder_gram"]], outputs[ti])   # no detach on teachers (loss.py:58-64)
    feat = feat / num_modal
    return total + loss_cfg["RC_Loss_weight"] * rc + loss_cfg["Feature_Loss_weight"] * feat


# --------------------------------------------------------------------------------------------
# state-dict template (names/shapes) without the reference: used by tests and bench cpu_baseline
# --------------------------------------------------------------------------------------------
def state_dict_template(cfg: OracleConfig, ffn_ratio=(3, 3, 2, 2), conv_exp=(3, 3, 2, 2), qkv_bias=True) -> Dict[str, Tensor]:
    """Zero tensors with the reference's key names / shapes."""
    sd: Dict[str, Tensor] = {}
    Z = lambda *s: torch.zeros(*s)
    M, p = cfg.M, cfg.patch_size
    ea = "encoder.encoder_attn."
    for m in range(M):
        sd[f"{ea}patch_embeds.{m}.proj.weight"] = Z(cfg.attn_base_ch, cfg.in_ch[m], p, p, p)
        sd[f"{ea}patch_embeds.{m}.proj.bias"] = Z(cfg.attn_base_ch)
    for L in range(4):
        C = cfg.attn_base_ch * 2 ** L
        pl = cfg.plans[L]
        for d in range(cfg.depths[L]):
            b = f"{ea}layers.{L}.blocks.{d}."
            n = pl["n"]
            sd[b + "attn.position_embedding.relative_position_bias_table"] = Z((2 * n[0] - 1) * (2 * n[1] - 1) * (2 * n[2] - 1), pl["heads"])
            sd[b + "attn.position_embedding.relative_position_index"] = relative_position_index(n)
            for m in range(M):
                sd[b + f"attn.input_norms.{m}.weight"] = Z(C)
                sd[b + f"attn.input_norms.{m}.bias"] = Z(C)
            for m in range(M):
                for j, co in enumerate((pl["ch_qk"], pl["ch_qk"], pl["ch_v"])):
                    sd[b + f"attn.qkv_proj.{m}.{j}.weight"] = Z(co, C, 1, 1, 1)
                    if qkv_bias:
                        sd[b + f"attn.qkv_proj.{m}.{j}.bias"] = Z(co)
            for m in range(M):
                sd[b + f"attn.mix_channels.{m}.weight"] = Z(C, pl["ch_v"], 1, 1, 1)
                sd[b + f"attn.mix_channels.{m}.bias"] = Z(C)
            for m in range(M):
                r = ffn_ratio[L]
                sd[b + f"ffns.{m}.linear1.weight"] = Z(C * r, C, 1, 1, 1)
                sd[b + f"ffns.{m}.linear1.bias"] = Z(C * r)
                sd[b + f"ffns.{m}.linear2.weight"] = Z(C, C * r, 1, 1, 1)
                sd[b + f"ffns.{m}.linear2.bias"] = Z(C)
            for m in range(M):
                sd[b + f"norms.{m}.weight"] = Z(C)
                sd[b + f"norms.{m}.bias"] = Z(C)
        if L < 3:
            for m in range(M):
                sd[f"{ea}layers.{L}.downs.{m}.reduction.weight"] = Z(2 * C, 8 * C, 1, 1, 1)
                sd[f"{ea}layers.{L}.downs.{m}.norm.weight"] = Z(8 * C)
                sd[f"{ea}layers.{L}.downs.{m}.norm.bias"] = Z(8 * C)

    def jlc_keys(pre, C, G, depth, r):
        for d in range(depth):
            for i, k in enumerate((1, 3, 5)):
                sd[f"{pre}{d}.spatial_convs.{i}.0.weight"] = Z(C, C // G, k, k, k)
                sd[f"{pre}{d}.spatial_convs.{i}.0.bias"] = Z(C)
            sd[f"{pre}{d}.channel_conv.1.weight"] = Z(C * r, C, 1, 1, 1)
            sd[f"{pre}{d}.channel_conv.1.bias"] = Z(C * r)
            sd[f"{pre}{d}.channel_conv.3.weight"] = Z(C, C * r, 1, 1, 1)
            sd[f"{pre}{d}.channel_conv.3.bias"] = Z(C)

    ec = "encoder.encoder_conv."
    cin = sum(cfg.in_ch)
    for L in range(4):
        C = cfg.base_ch * 2 ** L
        k = 2 * (p if L == 0 else 2) - 1
        sd[f"{ec}down{L + 1}.down.weight"] = Z(C, cin, k, k, k)
        sd[f"{ec}down{L + 1}.down.bias"] = Z(C)
        cin = C
    for L in range(4):
        jlc_keys(f"{ec}layer{L + 1}.", cfg.base_ch * 2 ** L, cfg.groups[L], cfg.conv_depths[L], conv_exp[L])
    for L in range(4):
        sd[f"encoder.attn2conv_{L + 1}.0.weight"] = Z(cfg.base_ch * 2 ** L, cfg.attn_base_ch * 2 ** L * M, 1, 1, 1)
        sd[f"encoder.attn2conv_{L + 1}.0.bias"] = Z(cfg.base_ch * 2 ** L)

    def trunk(pre):
        for L in (3, 2, 1):
            C = cfg.base_ch * 2 ** (L - 1)
            sd[f"{pre}layer_up{L}.up.weight"] = Z(2 * C, C, 2, 2, 2)
            sd[f"{pre}layer_up{L}.up.bias"] = Z(C)
        for L in (1, 2, 3):
            jlc_keys(f"{pre}layer{L}.", cfg.base_ch * 2 ** (L - 1), cfg.groups[L - 1], cfg.conv_depths[L - 1], conv_exp[L - 1])

    trunk("decoder.")
    sd["decoder.out_conv1.0.weight"] = Z(p ** 3 * cfg.n_classes, cfg.base_ch, 3, 3, 3)
    sd["decoder.out_conv1.0.bias"] = Z(p ** 3 * cfg.n_classes)
    if cfg.deep_supervision:
        for L in (2, 3, 4):
            sd[f"decoder.out_conv{L}.weight"] = Z(cfg.n_classes, cfg.base_ch * 2 ** (L - 1), 1, 1, 1)
            sd[f"decoder.out_conv{L}.bias"] = Z(cfg.n_classes)
    for m in range(M):
        pre = f"rc_decoders.{m}."
        for L in (4, 3, 2, 1):
            C = cfg.base_ch * 2 ** (L - 1)
            sd[f"{pre}enc2rc_{L}.0.weight"] = Z(C, (cfg.attn_base_ch + cfg.base_ch) * 2 ** (L - 1), 1, 1, 1)
            sd[f"{pre}enc2rc_{L}.0.bias"] = Z(C)
        trunk(pre)
        sd[f"{pre}out_conv.0.weight"] = Z(p ** 3 * cfg.in_ch[m], cfg.base_ch, 3, 3, 3)
        sd[f"{pre}out_conv.0.bias"] = Z(p ** 3 * cfg.in_ch[m])
    return sd

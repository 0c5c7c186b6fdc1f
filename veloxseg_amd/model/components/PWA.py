"""Paired-Window Attention blocks (reference: model/components/PWA.py).

Only the classes the VeloxSeg graph instantiates are provided (MultiModal_Paired_Windows_Attention,
Paired_Windows_TransformerBlock, Transformer_BasicLayer); the reference's Cross_Channel_Attention is dead code.
Geometry (window scales, channel split) is planned on the host once per layer and handed to the HIP
kernels as a VxPwaPlan (include/veloxseg_hip.h).
"""
from math import ceil
from typing import List, Sequence

import torch
from torch import nn

from ... import _hip as H
from ... import functional as VF
from .attention_utils import FFN, LayerNorm, PatchMerging, PositionalEmbedding
from .common_function import ParamConv3d


def plan_windows(input_size, min_big, min_small, scale_factor, num_heads, min_dim_head, channels):
    """Window scales `while (bw <= input).any()` and channel split (PWA.py:56-86)."""
    big, small = [], []
    bw, sw = [int(v) for v in min_big], [int(v) for v in min_small]
    while any(b <= g for b, g in zip(bw, input_size)):
        big.append(list(bw))
        small.append(list(sw))
        bw = [b * scale_factor for b in bw]
        sw = [s * scale_factor for s in sw]
    if not big:
        raise ValueError(f"PWA: window {list(min_big)} is larger than the token grid {list(input_size)} on every axis")
    need = len(big) * num_heads * min_dim_head
    ch_qk = need
    ch_v = ceil(channels / need) * need
    n = [min_big[k] // min_small[k] for k in range(3)]
    nwin = []
    for b in big:
        if any(g % x or g // x == 0 for g, x in zip(input_size, b)):
            raise ValueError(f"PWA: window {b} does not tile the token grid {list(input_size)} "
                             "(the reference fails inside einops.rearrange for the same configuration)")
        nwin.append([g // x for g, x in zip(input_size, b)])
    return dict(big=big, small=small, n=n, nwin=nwin, ch_qk=ch_qk, ch_v=ch_v)


class MultiModal_Paired_Windows_Attention(nn.Module):
    """LN -> q/k/v 1x1 -> window gather (max-pool + partition) -> multi-modal window attention with relative bias
    -> per-window trilinear scatter -> 1x1 mix -> x + Drop(.)   (PWA.py:246-379)."""

    def __init__(self, input_size: Sequence[int], in_channels: Sequence[int], min_big_window_size=(3, 3, 3), min_small_window_size=(1, 1, 1),
                 scale_factor: int = 2, num_heads: int = 1, min_dim_head: int = 4, qkv_bias: bool = True, attn_drop: float = 0.1,
                 proj_drop: float = 0.1, norm_layer=LayerNorm, dim: int = 3, use_pos_embed: bool = True):
        super().__init__()
        if dim != 3 or not use_pos_embed or num_heads < 1:
            raise NotImplementedError("veloxseg_amd PWA: 3-D, positional bias on, num_heads >= 1")
        self.input_size = list(input_size)
        self.in_channels = list(in_channels)
        self.num_modalities = len(in_channels)
        self.num_heads, self.min_dim_head, self.dim = num_heads, min_dim_head, dim
        self.mid_channels = max(in_channels)
        g = plan_windows(self.input_size, min_big_window_size, min_small_window_size, scale_factor, num_heads, min_dim_head, self.mid_channels)
        self.big_window_size, self.small_window_size = g["big"], g["small"]
        self.n_hwd = g["n"]
        self.num_bswin = len(g["big"])
        self.channels_qk, self.channels_v = g["ch_qk"], g["ch_v"]
        self.c_qk = self.channels_qk // (self.num_bswin * num_heads)
        self.c_v = self.channels_v // (self.num_bswin * num_heads)
        self.plan = H.make_plan(self.input_size, self.n_hwd, num_heads, g["small"], g["nwin"])
        self.position_embedding = PositionalEmbedding(dim=dim, num_heads=num_heads, window_size=self.n_hwd)
        self.attn_drop, self.proj_drop = attn_drop, proj_drop
        input_norms, qkv_proj, mix_channels, dropout_attns = [], [], [], []
        for m in range(self.num_modalities):
            input_norms.append(norm_layer(self.in_channels[m], data_format="channels_first", dim=dim))
            qkv_proj.append(nn.ModuleList([ParamConv3d(self.in_channels[m], self.channels_qk, kernel_size=1, bias=qkv_bias),
                                           ParamConv3d(self.in_channels[m], self.channels_qk, kernel_size=1, bias=qkv_bias),
                                           ParamConv3d(self.in_channels[m], self.channels_v, kernel_size=1, bias=qkv_bias)]))
            mix_channels.append(ParamConv3d(self.channels_v, self.in_channels[m], kernel_size=1))
            dropout_attns.append(nn.Dropout(proj_drop))
        self.input_norms = nn.ModuleList(input_norms)
        self.qkv_proj = nn.ModuleList(qkv_proj)
        self.mix_channels = nn.ModuleList(mix_channels)
        self.dropout_attns = nn.ModuleList(dropout_attns)
        self.site_attn = VF.new_dropout_site()
        self.sites_proj = [VF.new_dropout_site() for _ in range(self.num_modalities)]
        self._fused_ok = {}

    def _qkv(self, m: int, x):
        xn = self.input_norms[m](x)                      # LN once (the reference evaluates the same LN three times)
        cm = VF.cpp_node("conv") if xn.is_cuda else None
        if cm is not None:                               # one autograd node for the three projections: their input gradients accumulate in the kernels
            pq, pk, pv = self.qkv_proj[m]
            return list(cm.qkv(xn, pq.weight, pq.bias, pk.weight, pk.bias, pv.weight, pv.bias))
        return [self.qkv_proj[m][j](xn) for j in range(3)]

    def _post(self, m: int, x, s, residual_scale: float, tail):
        p = self.proj_drop if self.training else 0.0
        cm = VF.cpp_node("conv") if s.is_cuda else None
        mc = self.mix_channels[m]
        if cm is not None and cm.pw_res_ok(s, mc.weight):      # mix conv + residual + dropout: one launch, one autograd node
            y = cm.pw_res(s, mc.weight, mc.bias, x, float(residual_scale), float(p), int(self.sites_proj[m]), VF.rs_ptr(s.device, p))
        else:
            mix = mc(s)
            y = VF.residual_dropout(x, mix, residual_scale, p, self.sites_proj[m])
        return tail(m, y) if tail is not None else y

    def _fused_pre_ok(self, x):
        if not (VF.USE_PWA_FUSED and x.is_cuda and x.dtype == torch.float32 and 1 <= self.num_modalities <= 4 and len(set(self.in_channels)) == 1):
            return False
        if any((c.bias is None) != (self.qkv_proj[0][0].bias is None) for pr in self.qkv_proj for c in pr):
            return False
        key = ("pre", int(x[0, 0].numel()))
        ok = self._fused_ok.get(key)
        if ok is None:
            ok = self._fused_ok[key] = VF.ln_pw_ok(self.in_channels[0], [self.channels_qk, self.channels_qk, self.channels_v], key[1])
        return ok

    def _fused_post_ok(self, x, ffn):
        """the whole tail of the block (mix conv + residual, LN, FFN) as one launch: where csrc/mlp.hip does not cover the FFN (the 8^3 / 4^3 levels)"""
        if ffn is None or not (VF.USE_PWA_FUSED and x.is_cuda and len(set(self.in_channels)) == 1) or any(mc.bias is None for mc in self.mix_channels):
            return False
        norms, ffns = ffn
        C, V = self.in_channels[0], int(x[0, 0].numel())
        R = int(ffns[0].linear1.weight.shape[0])
        key = ("post", V, R)
        ok = self._fused_ok.get(key)
        if ok is None:
            ok = self._fused_ok[key] = (VF.pwa_post_ok(C, self.channels_v, R, V) and (not bool(H.query("vx_mlp_supported", C, R, V)) or C >= VF.TILE_MIN_C)
                                        and all(f.p == ffns[0].p for f in ffns))
        return ok

    def forward(self, inputs: List[torch.Tensor], residual_scale: float = 1.0, tail=None, ffn=None) -> List[torch.Tensor]:
        """returns residual_scale * x_m + Drop(mix(attention)) ; the transformer block passes 2.0 (double residual).
        Everything except the joint attention is per modality and independent.  Default (functional.USE_PWA_FUSED): LN + q / k / v of ALL modalities
        are one launch (csrc/pwa_fused.hip), and so is -- given `ffn` = (norms, ffns) of the enclosing block, at the levels csrc/mlp.hip does not
        cover -- the whole tail mix conv + residual + LN + FFN.  Otherwise, with functional.MODALITY_STREAMS, the M modalities run on forked HIP
        streams before and after the attention (`tail(m, y)`, e.g. the block's FFN, rides on the same branch)."""
        M = self.num_modalities
        assert len(inputs) == M, f"The number of modalities should be {M}, but got {len(inputs)}"
        if self._fused_pre_ok(inputs[0]):
            parts, res = VF.pwa_pre(inputs, self.input_norms, self.qkv_proj)      # res[m]: x_m passed through (its gradient is added inside the pre-backward kernel)
            qkv = [t for p_ in parts for t in p_]
            scat = VF.pwa_core(self.position_embedding.relative_position_bias_table, self.plan, self.c_qk, self.c_v, qkv,
                               self.attn_drop if self.training else 0.0, self.site_attn)
            if self._fused_post_ok(inputs[0], ffn):
                norms, ffns = ffn
                return VF.pwa_post(scat, res, self.mix_channels, norms, ffns, residual_scale, self.proj_drop if self.training else 0.0,
                                   ffns[0].p if self.training else 0.0, self.sites_proj)
            par = VF.MODALITY_STREAMS and VF.BRANCH_STREAMS and M > 1
            if par:
                return VF.run_branches([(lambda m=m: self._post(m, res[m], scat[m], residual_scale, tail)) for m in range(M)], inputs[0].device,
                                       tag="modalities", uses=[[res[m], scat[m]] for m in range(M)])
            return [self._post(m, res[m], scat[m], residual_scale, tail) for m in range(M)]
        par = VF.MODALITY_STREAMS and VF.BRANCH_STREAMS and M > 1 and inputs[0].is_cuda
        if par:
            parts = VF.run_branches([(lambda m=m: self._qkv(m, inputs[m])) for m in range(M)], inputs[0].device, tag="modalities",
                                    uses=[[inputs[m]] for m in range(M)])
        else:
            parts = [self._qkv(m, inputs[m]) for m in range(M)]
        qkv = [t for p in parts for t in p]
        scat = VF.pwa_core(self.position_embedding.relative_position_bias_table, self.plan, self.c_qk, self.c_v, qkv,
                           self.attn_drop if self.training else 0.0, self.site_attn)
        if par:
            return VF.run_branches([(lambda m=m: self._post(m, inputs[m], scat[m], residual_scale, tail)) for m in range(M)], inputs[0].device,
                                   tag="modalities", uses=[[inputs[m], scat[m]] for m in range(M)])
        return [self._post(m, inputs[m], scat[m], residual_scale, tail) for m in range(M)]


class Paired_Windows_TransformerBlock(nn.Module):
    """y = x + attn(x) (attn already contains +x: double residual, PWA.py:377,436); z = y + FFN(LN(y)) (:437)."""

    def __init__(self, input_size, in_channels, min_big_window_size=(3, 3, 3), min_small_window_size=(1, 1, 1), scale_factor=2, num_heads=1,
                 min_dim_head=4, attn_drop=0.1, proj_drop=0.1, drop_path=0.0, ffn_expansion_ratio=4, act_layer="GELU", norm_layer=LayerNorm,
                 qkv_bias=True, dim=3):
        super().__init__()
        if drop_path and drop_path > 0:
            raise NotImplementedError("drop_path > 0 is not used by any shipped config")
        self.input_size, self.in_channels = input_size, in_channels
        self.num_modalities = len(in_channels)
        self.attn = MultiModal_Paired_Windows_Attention(input_size=input_size, in_channels=in_channels, min_big_window_size=min_big_window_size,
                                                        min_small_window_size=min_small_window_size, scale_factor=scale_factor, num_heads=num_heads,
                                                        min_dim_head=min_dim_head, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=proj_drop,
                                                        norm_layer=norm_layer, dim=dim, use_pos_embed=True)
        self.drop_path = nn.Identity()
        self.ffns = nn.ModuleList()
        self.norms = nn.ModuleList()
        for m in range(self.num_modalities):
            self.ffns.append(FFN(in_channels[m], expansion_ratio=ffn_expansion_ratio, dropout_rate=proj_drop, act=act_layer, dim=dim))
            self.norms.append(norm_layer(in_channels[m], data_format="channels_first", dim=dim))

    def forward(self, xs):
        if VF.USE_COMPOSITE:
            tail = lambda m, y: VF.ffn_tail(y, self.norms[m], self.ffns[m], self.ffns[m].p if self.training else 0.0)      # noqa: E731
        else:
            tail = lambda m, y: self.ffns[m](self.norms[m](y), residual=y)      # noqa: E731
        return self.attn(xs, residual_scale=2.0, tail=tail, ffn=(self.norms, self.ffns))


class _FeatureList(list):
    """a list of per-modality features that can carry the aliases meant for the decoders (functional.fan_out)"""


class Transformer_BasicLayer(nn.Module):
    """depth x block, then optional PatchMerging per modality (PWA.py:444-511)."""

    def __init__(self, input_size, in_channels, depth=2, min_big_window_size=(3, 3, 3), min_small_window_size=(1, 1, 1), scale_factor=2,
                 num_heads=1, min_dim_head=4, attn_drop=0.1, proj_drop=0.1, drop_path=0, ffn_expansion_ratio=4, act_layer="GELU",
                 norm_layer=LayerNorm, qkv_bias=True, do_downsample=True, dim=3):
        super().__init__()
        self.num_modalities = len(in_channels)
        self.blocks = nn.ModuleList([
            Paired_Windows_TransformerBlock(input_size=input_size, in_channels=in_channels, min_big_window_size=min_big_window_size,
                                            min_small_window_size=min_small_window_size, scale_factor=scale_factor, num_heads=num_heads,
                                            min_dim_head=min_dim_head, attn_drop=attn_drop, proj_drop=proj_drop,
                                            drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                            ffn_expansion_ratio=ffn_expansion_ratio, act_layer=act_layer, norm_layer=norm_layer,
                                            qkv_bias=qkv_bias, dim=dim)
            for i in range(depth)])
        self.downs = None
        if do_downsample:
            self.downs = nn.ModuleList([PatchMerging(in_ch=in_channels[m], norm_layer=norm_layer, dim=dim) for m in range(self.num_modalities)])

    def forward(self, xs):
        for blk in self.blocks:
            xs = blk(xs)
        if self.training and torch.is_grad_enabled():
            # the level's features have three consumers (the modal mixer, the decoders, the next level's patch merge; two at the last level): one alias each, so that
            # their gradients are summed in one launch (functional.fan_out) -- `for_decoders` travels with the returned list (model/Encoder.py)
            al = VF.fan_out(xs, 2 if self.downs is None else 3)
            out = _FeatureList(a[0] for a in al)
            out.for_decoders = [a[1] for a in al]
            if self.downs is None:
                return out, None
            xs = [a[2] for a in al]
        else:
            out = xs
            if self.downs is None:
                return xs, None
        M = self.num_modalities
        if (VF.USE_PWA_FUSED and xs[0].is_cuda and 1 <= M <= 4 and len({d.in_ch for d in self.downs}) == 1
                and all(v % 2 == 0 for v in xs[0].shape[2:]) and VF.ln_pw_ok(8 * self.downs[0].in_ch, [2 * self.downs[0].in_ch], xs[0][0, 0].numel() // 8, True)):
            return out, VF.patch_merge_all(xs, self.downs)          # gather + LN(8C) + reduction of every modality: one launch
        if VF.MODALITY_STREAMS >= 2 and VF.BRANCH_STREAMS and M > 1 and xs[0].is_cuda:       # PatchMerging is per modality too
            down = VF.run_branches([(lambda m=m: self.downs[m](xs[m])) for m in range(M)], xs[0].device, tag="modalities", uses=[[xs[m]] for m in range(M)])
        else:
            down = [self.downs[m](xs[m]) for m in range(M)]
        return out, down

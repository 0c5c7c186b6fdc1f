"""Dual-branch encoder (reference: model/Encoder.py)."""
from typing import Sequence

import torch
from torch import nn

from .. import functional as VF
from .components.attention_utils import LayerNorm
from .components.common_function import InstanceNormMarker, ParamConv3d
from .components.conv_blocks import DownConv, JLCLayer
from .components.PWA import Transformer_BasicLayer


class PatchEmbed(nn.Module):
    """MONAI PatchEmbed stand-in: proj = Conv3d(k = s = patch), no norm (Encoder.py:150-156)."""

    def __init__(self, patch_size, in_chans, embed_dim, norm_layer=None, spatial_dims=3):
        super().__init__()
        if norm_layer is not None or spatial_dims != 3:
            raise NotImplementedError("patch_norm=True / 2-D are not used by any shipped config")
        self.patch_size = (patch_size,) * 3
        self.proj = ParamConv3d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = None

    def forward(self, x):
        if any(s % p for s, p in zip(x.shape[2:], self.patch_size)):
            raise ValueError(f"input {tuple(x.shape[2:])} must be a multiple of patch_size {self.patch_size}")
        return self.proj(x)


class Conv_Encoder(nn.Module):
    """DownConv + JLC x4 (Encoder.py:13-85)."""

    def __init__(self, patch_size=4, in_ch=1, base_ch=16, depths=(1, 1, 1, 1), kernel_sizes=(1, 3, 5), min_dim_group=(4, 8, 8, 16),
                 expansion_factor=(3, 3, 2, 2), dropout=0.0, spatial_dim=3):
        super().__init__()
        self.down1 = DownConv(in_ch, base_ch, patch_size=patch_size, dim=spatial_dim)
        self.down2 = DownConv(base_ch, base_ch * 2, patch_size=2, dim=spatial_dim)
        self.down3 = DownConv(base_ch * 2, base_ch * 4, patch_size=2, dim=spatial_dim)
        self.down4 = DownConv(base_ch * 4, base_ch * 8, patch_size=2, dim=spatial_dim)
        groups = [base_ch * 2 ** i // min_dim_group[i] for i in range(4)]
        for i in range(4):
            setattr(self, f"layer{i + 1}", JLCLayer(base_ch * 2 ** i, depths[i], kernel_sizes, groups[i], expansion_factor[i],
                                                     dropout=dropout, spatial_dim=spatial_dim))

    def forward(self, x):
        feats = []
        for i in range(4):
            x = getattr(self, f"layer{i + 1}")(getattr(self, f"down{i + 1}")(x))
            feats.append(x)
        return tuple(feats)


class Transformer_Encoder(nn.Module):
    """per-modality PatchEmbed -> 4 PWA layers with PatchMerging (Encoder.py:88-204)."""

    def __init__(self, input_size, patch_size, in_channels, embed_dim=16, depths=(2, 2, 2, 2),
                 min_big_window_sizes=((3, 3, 3), (6, 6, 6), (3, 3, 3), (3, 3, 3)), min_small_window_sizes=((1, 1, 1),) * 4,
                 scale_factors=(2, 2, 2, 2), num_heads=(1, 2, 2, 4), min_dim_head=(4, 8, 8, 16), ffn_expansion_ratio=(3, 3, 2, 2),
                 attn_drop=0.1, proj_drop=0.1, drop_path=0, act_layer="GELU", norm_layer=LayerNorm, patch_norm=False, qkv_bias=True, spatial_dim=3):
        super().__init__()
        self.in_channels = list(in_channels)
        self.num_modalities = len(in_channels)
        self.num_layers = len(depths)
        self.patch_size = patch_size
        self.patch_embeds = nn.ModuleList([PatchEmbed(patch_size=patch_size, in_chans=self.in_channels[m], embed_dim=embed_dim,
                                                      norm_layer=norm_layer if patch_norm else None, spatial_dims=spatial_dim)
                                           for m in range(self.num_modalities)])
        self.pos_drop = nn.Dropout(p=proj_drop)
        self.p_pos = proj_drop
        self.sites_pos = [VF.new_dropout_site() for _ in range(self.num_modalities)]
        if drop_path:
            raise NotImplementedError("drop_path > 0 is not used by any shipped config")
        self.layers = nn.ModuleList()
        grid = [int(s) // patch_size for s in input_size]
        for i in range(self.num_layers):
            self.layers.append(Transformer_BasicLayer(
                input_size=list(grid), in_channels=[int(embed_dim * 2 ** i)] * self.num_modalities, depth=depths[i],
                min_big_window_size=min_big_window_sizes[i], min_small_window_size=min_small_window_sizes[i], scale_factor=scale_factors[i],
                num_heads=num_heads[i], min_dim_head=min_dim_head[i], attn_drop=attn_drop, proj_drop=proj_drop, drop_path=0.0,
                ffn_expansion_ratio=ffn_expansion_ratio[i], act_layer=act_layer, norm_layer=norm_layer, qkv_bias=qkv_bias,
                do_downsample=i < self.num_layers - 1, dim=spatial_dim))
            grid = [g // 2 for g in grid]

    def embed(self, xs):
        xs = torch.chunk(xs, self.num_modalities, dim=1)            # Encoder.py:192
        p = self.p_pos if self.training else 0.0
        def one(m):
            e = self.patch_embeds[m](xs[m])          # a channel slice: the patch-embedding kernels read it in place (vx_patchify_bs)
            return VF.residual_dropout(None, e, 0.0, p, self.sites_pos[m]) if p > 0 else e

        M = self.num_modalities
        if VF.MODALITY_STREAMS >= 2 and VF.BRANCH_STREAMS and M > 1 and xs[0].is_cuda:
            return VF.run_branches([(lambda m=m: one(m)) for m in range(M)], xs[0].device, tag="modalities", uses=[[xs[m]] for m in range(M)])
        return [one(m) for m in range(M)]

    def forward(self, xs):
        cur = self.embed(xs)
        feats = []
        for i, layer in enumerate(self.layers):
            attn, cur = layer(cur)
            feats.append(attn)
        return tuple(feats)


class Encoder(nn.Module):
    """PWA branch + conv branch, fused per level by a 1x1 modal mixer and an add (Encoder.py:207-367)."""

    def __init__(self, input_size, patch_size, in_ch, base_ch=16, conv_depths=(1, 1, 1, 1), kernel_sizes=(1, 3, 5), min_dim_group=(4, 8, 8, 16),
                 conv_expansion_factor=(4, 4, 4, 4), attn_base_ch=16, depths=(2, 2, 2, 2),
                 min_big_window_sizes=((3, 3, 3), (6, 6, 6), (3, 3, 3), (3, 3, 3)), min_small_window_sizes=((1, 1, 1),) * 4,
                 min_dim_head=(4, 8, 8, 16), scale_factors=(2, 2, 2, 2), num_heads=(1, 2, 4, 8), attn_drop=0.1, proj_drop=0.1, drop_path=0,
                 ffn_expansion_ratio=(4, 4, 4, 4), act_layer="GELU", norm_layer=LayerNorm, patch_norm=False, qkv_bias=True, conv_drop=0.0, spatial_dim=3):
        super().__init__()
        self.in_channels = list(in_ch)
        self.num_modalities = len(in_ch)
        self.encoder_attn = Transformer_Encoder(input_size=input_size, patch_size=patch_size, in_channels=in_ch, embed_dim=attn_base_ch, depths=depths,
                                                min_big_window_sizes=min_big_window_sizes, min_small_window_sizes=min_small_window_sizes,
                                                scale_factors=scale_factors, num_heads=num_heads, min_dim_head=min_dim_head, attn_drop=attn_drop,
                                                proj_drop=proj_drop, drop_path=drop_path, ffn_expansion_ratio=ffn_expansion_ratio, act_layer=act_layer,
                                                norm_layer=norm_layer, patch_norm=patch_norm, qkv_bias=qkv_bias, spatial_dim=spatial_dim)
        self.encoder_conv = Conv_Encoder(patch_size=patch_size, in_ch=sum(in_ch), base_ch=base_ch, depths=conv_depths, kernel_sizes=kernel_sizes,
                                         min_dim_group=min_dim_group, expansion_factor=conv_expansion_factor, dropout=conv_drop, spatial_dim=spatial_dim)
        M = self.num_modalities
        self._on_level_inputs = None        # engine hook: called with (level 2..4, the tensors level L-1 hands to level L) during a training forward
        for i in range(4):
            setattr(self, f"attn2conv_{i + 1}", nn.Sequential(ParamConv3d(attn_base_ch * 2 ** i * M, base_ch * 2 ** i, 1, 1), InstanceNormMarker(base_ch * 2 ** i)))

    def _mix(self, level: int, attn_feats):
        """1x1 conv over the channel-concatenated modalities (the concat is done inside the kernel for M = 2)."""
        conv = getattr(self, f"attn2conv_{level + 1}")[0]
        if len(attn_feats) == 1:
            return conv(attn_feats[0])
        if len(attn_feats) == 2:
            return conv(attn_feats[0], x2=attn_feats[1])
        return conv(torch.cat(list(attn_feats), dim=1))

    def forward(self, x):
        """The PWA chain (transformer levels 1..4) and the conv chain only meet at the per-level mixers, conv level L needing attn_L
        (Encoder.py:351-360).  With functional.BRANCH_STREAMS the conv chain runs on a second HIP stream, one level behind the
        transformer: conv level L overlaps transformer level L+1, forward and (autograd replays nodes on their forward stream) backward."""
        ta = self.encoder_attn
        cur_feats = ta.embed(x)
        side = VF.side_stream(x.device, "encoder_conv") if VF.BRANCH_STREAMS else None
        main = torch.cuda.current_stream(x.device) if side is not None else None
        if side is not None:
            side.wait_stream(main)
            x.record_stream(side)            # read by down1 on the side stream, forward and (weight gradient) backward
        attn, encs = [], []
        prev = x
        for i in range(4):
            if i > 0 and self._on_level_inputs is not None and self.training:
                self._on_level_inputs(i + 1, list(cur_feats) + [prev])       # level i+1 consumes the merged tokens and the conv feature of level i
            # the strided DownConv of this level only needs the previous conv feature: it is queued on the conv-chain stream BEFORE that stream waits for
            # the transformer level, so it overlaps PWA level i; and because the mixer is then the LAST node of the pair in the forward order, autograd
            # runs its backward FIRST -- the attention gradients leave the conv chain before the DownConv's input / weight gradients are computed
            if side is not None:
                ctx = torch.cuda.stream(side)
                ctx.__enter__()
            try:
                d_raw = getattr(self.encoder_conv, f"down{i + 1}").raw(prev)
            finally:
                if side is not None:
                    ctx.__exit__(None, None, None)
            a_i, cur_feats = ta.layers[i](cur_feats)
            attn.append(getattr(a_i, "for_decoders", a_i))      # (training: the decoders get aliases of their own, functional.fan_out)
            if side is not None:
                side.wait_stream(main)
                for t_ in a_i:
                    t_.record_stream(side)   # the mixer reads them on the side stream
                ctx = torch.cuda.stream(side)
                ctx.__enter__()
            try:
                a_raw = self._mix(i, a_i)
                fused = VF.instnorm_sum([d_raw, a_raw])                  # IN(down) + IN(mix)  (Encoder.py:351-360)
                prev = getattr(self.encoder_conv, f"layer{i + 1}")(fused)
                prev_dec = prev
                if i < 3 and self.training and torch.is_grad_enabled():
                    (prev_dec, prev), = VF.fan_out([prev], 2)              # consumers: the decoders and the next level's DownConv
            finally:
                if side is not None:
                    ctx.__exit__(None, None, None)
            encs.append(prev_dec)
        if side is not None:
            main.wait_stream(side)
            for e in encs:
                e.record_stream(main)
        if self.training:
            return [list(a) for a in attn], encs
        return tuple(encs)

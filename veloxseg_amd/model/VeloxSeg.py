"""VeloxSeg top module -- drop-in for the reference's model/VeloxSeg.py:16 (same constructor kwargs,
attribute tree => same state_dict keys, same train / eval return conventions), computed by HIP kernels."""
from typing import Sequence, Union

import torch
from torch import nn

from .. import functional as VF
from .components.attention_utils import LayerNorm
from .components.initialization import InitWeights_He
from .Decoder import RC_Decoder, Seg_Decoder
from .Encoder import Encoder


class VeloxSeg(nn.Module):
    def __init__(self, input_size: Sequence[int], patch_size: int, in_ch: Sequence[int], n_classes: int = 2, base_ch: int = 16,
                 conv_depths: Sequence[int] = (1, 1, 1, 1), kernel_sizes: Sequence[int] = (1, 3, 5), min_dim_group: Sequence[int] = (4, 8, 8, 16),
                 conv_expansion_factor: Sequence[int] = (3, 3, 2, 2), attn_base_ch: int = 16, depths: Sequence[int] = (2, 2, 2, 2),
                 min_big_window_sizes=((3, 3, 3), (6, 6, 6), (3, 3, 3), (3, 3, 3)), min_small_window_sizes=((1, 1, 1),) * 4,
                 min_dim_head: Sequence[int] = (4, 8, 8, 16), scale_factors: Sequence[int] = (2, 2, 2, 2), num_heads: Sequence[int] = (1, 2, 2, 4),
                 attn_drop: float = 0.1, proj_drop: float = 0.1, drop_path: float = 0, ffn_expansion_ratio: Sequence[int] = (3, 3, 2, 2),
                 act_layer: str = "GELU", norm_layer=LayerNorm, patch_norm: bool = False, qkv_bias: bool = True, conv_drop: float = 0.0,
                 deep_supervision: bool = True, spatial_dim: int = 3):
        super().__init__()
        if spatial_dim != 3:
            raise NotImplementedError("veloxseg_amd implements the 3-D network (all shipped configs)")
        VF.reset_dropout_sites()
        self._on_encoder_outputs = None
        # True (set by engine.TrainEngine): the training forward returns the deep-supervision heads 1.. on THEIR OWN grids and veloxseg_amd.utils.loss.Loss
        # interpolates them inside the loss kernels (csrc/loss_ds.hip) -- same loss and gradients, the three (B, ncls, S^3) tensors are never written.
        # False (default) = the reference's output list: every head up-sampled to the input size (VeloxSeg.py:202).
        self.ds_fused = False
        self.size = list(input_size)
        self.spatial_dim = spatial_dim
        self.patch_size = patch_size
        self.in_ch = list(in_ch)
        self.n_classes = n_classes
        self.num_modalities = len(in_ch)
        self.encoder = Encoder(input_size=input_size, patch_size=patch_size, in_ch=in_ch, base_ch=base_ch, conv_depths=conv_depths,
                               kernel_sizes=kernel_sizes, min_dim_group=min_dim_group, conv_expansion_factor=conv_expansion_factor,
                               attn_base_ch=attn_base_ch, depths=depths, min_big_window_sizes=min_big_window_sizes,
                               min_small_window_sizes=min_small_window_sizes, min_dim_head=min_dim_head, scale_factors=scale_factors,
                               num_heads=num_heads, attn_drop=attn_drop, proj_drop=proj_drop, drop_path=drop_path,
                               ffn_expansion_ratio=ffn_expansion_ratio, act_layer=act_layer, norm_layer=norm_layer, patch_norm=patch_norm,
                               qkv_bias=qkv_bias, conv_drop=conv_drop, spatial_dim=spatial_dim)
        self.decoder = Seg_Decoder(patch_size=patch_size, base_ch=base_ch, out_ch=n_classes, depths=conv_depths, kernel_sizes=kernel_sizes,
                                   min_dim_group=min_dim_group, expansion_factor=conv_expansion_factor, dropout=conv_drop,
                                   deep_supervision=deep_supervision, spatial_dim=spatial_dim)
        self.rc_decoders = nn.ModuleList([
            RC_Decoder(in_channel=in_ch[i], enc_channel=attn_base_ch + base_ch, dec_channel=base_ch, patch_size=patch_size, depths=conv_depths,
                       kernel_sizes=kernel_sizes, min_dim_group=min_dim_group, expansion_factor=conv_expansion_factor, spatial_dim=spatial_dim,
                       dropout=conv_drop) for i in range(len(in_ch))])
        self.init_weights()

    def init_weights(self):
        self.apply(InitWeights_He(neg_slope=1e-2))

    def scale_prediction(self, pred):
        """trilinear, align_corners=True, to self.size (VeloxSeg.py:177-184)."""
        return VF.upsample_trilinear(pred, self.size)

    # Training-mode heads (VeloxSeg.py:199-221): Seg_Decoder + deep-supervision up-sampling (branch 0) and the M reconstruction
    # decoders (branches 1..M).  The branches only share their inputs, so they can run concurrently (functional.run_branches in the
    # eager path, one hipGraph per branch in engine.TrainEngine).
    @property
    def num_branches(self) -> int:
        return 1 + self.num_modalities

    def decode_branch(self, k: int, attn, encs):
        """branch 0 -> (pred_0..pred_3, dec_pram); branch m+1 -> (rc_m, rc_pram_m).  attn[L][m] / encs[L] as returned by the encoder."""
        if k == 0:
            pred, dec_pram = self.decoder(*encs)
            if self.ds_fused:
                return (self.scale_prediction(pred[0]),) + tuple(pred[1:]) + (dec_pram,)
            return tuple(self.scale_prediction(p) for p in pred) + (dec_pram,)
        m = k - 1
        return tuple(self.rc_decoders[m]([attn[L][m] for L in range(4)], encs))

    def assemble_train(self, branch_outs):
        """list of decode_branch results -> the reference's training output list [pred x4, rcs, dec_pram, rc_pram x M] (VeloxSeg.py:221)"""
        seg = list(branch_outs[0])
        rcs = [o[0] for o in branch_outs[1:]]
        rcs = rcs[0] if len(rcs) == 1 else torch.cat(rcs, dim=1)
        return seg[:-1] + [rcs, seg[-1]] + [o[1] for o in branch_outs[1:]]

    def decode_train(self, attn, encs):
        outs = VF.run_branches([(lambda k=k: self.decode_branch(k, attn, encs)) for k in range(self.num_branches)], encs[0].device,
                               uses=[[encs] if k == 0 else [encs, [attn[L][k - 1] for L in range(4)]] for k in range(self.num_branches)])
        return self.assemble_train(outs)

    def forward(self, x) -> Union[torch.Tensor, Sequence[torch.Tensor]]:
        if not x.is_cuda:
            raise RuntimeError("veloxseg_amd.VeloxSeg runs on MI355X only: move the model and the input to a cuda device "
                               "(there is deliberately no CPU fallback; the CPU oracle lives under oracle/ for tests)")
        x = x.contiguous()
        VF.ensure_streams(x.device, self.num_modalities)
        if self.training:
            VF.advance_rng(x.device)
            attn, encs = self.encoder(x)
            if self._on_encoder_outputs is not None:          # engine hook: lets the data-parallel step see where the decoders' backward ends
                self._on_encoder_outputs(attn, encs)
            return self.decode_train(attn, encs)
        encs = self.encoder(x)
        return self.decoder(*encs)

"""Segmentation (student) and reconstruction (teacher) decoders (reference: model/Decoder.py)."""
from torch import nn

from .. import functional as VF
from .components.common_function import InstanceNormMarker, ParamConv3d, get_pram_matrix
from .components.conv_blocks import JLCLayer, UpConv
from .components.superpixel import PixelShuffle


def _trunk_modules(mod, ch, depths, kernel_sizes, min_dim_group, expansion_factor, dropout, spatial_dim):
    mod.layer_up3 = UpConv(ch * 8, ch * 4, up_rate=2, dim=spatial_dim)
    mod.layer_up2 = UpConv(ch * 4, ch * 2, up_rate=2, dim=spatial_dim)
    mod.layer_up1 = UpConv(ch * 2, ch, up_rate=2, dim=spatial_dim)
    groups = [ch * 2 ** i // min_dim_group[i] for i in range(4)]
    mod.layer1 = JLCLayer(ch, depths[0], kernel_sizes, groups[0], expansion_factor[0], dropout=dropout, spatial_dim=spatial_dim)
    mod.layer2 = JLCLayer(ch * 2, depths[1], kernel_sizes, groups[1], expansion_factor[1], dropout=dropout, spatial_dim=spatial_dim)
    mod.layer3 = JLCLayer(ch * 4, depths[2], kernel_sizes, groups[2], expansion_factor[2], dropout=dropout, spatial_dim=spatial_dim)


def _trunk_forward(mod, e1, e2, e3, e4):
    up3 = mod.layer3(mod.layer_up3(e4, skip=e3))
    up2 = mod.layer2(mod.layer_up2(up3, skip=e2))
    up1 = mod.layer1(mod.layer_up1(up2, skip=e1))
    return up1, up2, up3


class RC_Decoder(nn.Module):
    """Teacher: enc2rc 1x1+IN on cat(attn_m, enc), UpConv+JLC trunk, 3^3 conv + PixelShuffle, Gram of up1 (Decoder.py:11-94)."""

    def __init__(self, in_channel, enc_channel, dec_channel, patch_size, depths=(1, 1, 1, 1), kernel_sizes=(1, 3, 5), min_dim_group=(4, 8, 8, 16),
                 expansion_factor=(3, 3, 2, 2), spatial_dim=3, dropout=0.0):
        super().__init__()
        for L in (4, 3, 2, 1):
            s = 2 ** (L - 1)
            setattr(self, f"enc2rc_{L}", nn.Sequential(ParamConv3d(enc_channel * s, dec_channel * s, 1, 1, 0), InstanceNormMarker(dec_channel * s)))
        _trunk_modules(self, dec_channel, depths, kernel_sizes, min_dim_group, expansion_factor, dropout, spatial_dim)
        self.patch_size = patch_size
        self.out_conv = nn.Sequential(ParamConv3d(dec_channel, patch_size ** 3 * in_channel, kernel_size=3, stride=1, padding=1),
                                      PixelShuffle(scale=patch_size, spatial_dim=spatial_dim))

    def forward(self, attn_feats, enc_feats):
        """attn_feats / enc_feats: 4 tensors each; the channel concat (VeloxSeg.py:209-214) happens inside the 1x1 kernel."""
        e = [VF.instnorm_sum([getattr(self, f"enc2rc_{L + 1}")[0](attn_feats[L], x2=enc_feats[L])]) for L in range(4)]
        up1, _, _ = _trunk_forward(self, *e)
        # head_bf16 (set by engine.TrainEngine for its staged passes in the bf16 storage mode): the full-resolution output as a bfloat16 tensor
        rc = self.out_conv[0](up1, pixel_shuffle=self.patch_size, out_bf16=self.training and getattr(self, "head_bf16", False))
        if self.training:
            return rc, get_pram_matrix(up1)
        return rc


class Seg_Decoder(nn.Module):
    """Student (Decoder.py:97-179)."""

    def __init__(self, patch_size, base_ch=32, out_ch=2, depths=(1, 1, 1, 1), kernel_sizes=(1, 3, 5), min_dim_group=(4, 8, 8, 16),
                 expansion_factor=(3, 3, 2, 2), dropout=0.0, deep_supervision=False, spatial_dim=3):
        super().__init__()
        self.deep_supervision = deep_supervision
        _trunk_modules(self, base_ch, depths, kernel_sizes, min_dim_group, expansion_factor, dropout, spatial_dim)
        self.patch_size = patch_size
        self.out_conv1 = nn.Sequential(ParamConv3d(base_ch, patch_size ** 3 * out_ch, kernel_size=3, stride=1, padding=1),
                                       PixelShuffle(scale=patch_size, spatial_dim=spatial_dim))
        if deep_supervision:
            self.out_conv2 = ParamConv3d(base_ch * 2, out_ch, 1, 1)
            self.out_conv3 = ParamConv3d(base_ch * 4, out_ch, 1, 1)
            self.out_conv4 = ParamConv3d(base_ch * 8, out_ch, 1, 1)

    def forward(self, enc1, enc2, enc3, enc4):
        up1, up2, up3 = _trunk_forward(self, enc1, enc2, enc3, enc4)
        out = self.out_conv1[0](up1, pixel_shuffle=self.patch_size, out_bf16=self.training and getattr(self, "head_bf16", False))
        if self.training:
            if self.deep_supervision:
                return [out, self.out_conv2(up2), self.out_conv3(up3), self.out_conv4(enc4)], get_pram_matrix(up1)
            return [out], get_pram_matrix(up1)
        return out

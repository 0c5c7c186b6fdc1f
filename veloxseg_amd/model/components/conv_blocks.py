"""DownConv / UpConv / JLC (reference: model/components/conv_blocks.py)."""
from torch import nn

from ... import functional as VF
from .common_function import InstanceNormMarker, Marker, ParamConv3d, ParamConvTranspose3d


class DownConv(nn.Module):
    """Conv3d(k=2p-1, stride p, pad p-1) + InstanceNorm (conv_blocks.py:4-21)."""

    def __init__(self, in_channels, out_channels, patch_size=2, groups=1, use_norm=True, dim=3):
        super().__init__()
        assert dim == 3 and use_norm
        self.down = ParamConv3d(in_channels, out_channels, kernel_size=2 * patch_size - 1, stride=patch_size, padding=patch_size - 1, groups=groups)
        self.norm = InstanceNormMarker(out_channels)

    def raw(self, x):
        """un-normalised conv output; callers fuse the IN with whatever is added next"""
        return self.down(x)

    def forward(self, x):
        return VF.instnorm_sum([self.down(x)])


class UpConv(nn.Module):
    """ConvTranspose3d(k2, s2) + InstanceNorm (conv_blocks.py:23-39)."""

    def __init__(self, in_channels, out_channels, up_rate=2, groups=1, dim=3):
        super().__init__()
        assert dim == 3 and up_rate == 2 and groups == 1
        self.up = ParamConvTranspose3d(in_channels, out_channels, kernel_size=up_rate, stride=up_rate, groups=groups)
        self.norm = InstanceNormMarker(out_channels)

    def forward(self, x, skip=None):
        """IN(convT(x)) (+ skip)"""
        return VF.instnorm_sum([self.up(x, feeds_instnorm=True)], act=False, res=skip)


class JLC(nn.Module):
    """x + sum_k GELU(IN(gconv_k(x))), then + Drop(1x1(GELU(1x1(IN(.)))))  (conv_blocks.py:41-75)."""

    def __init__(self, in_channels, kernel_sizes=(1, 3, 5), groups=1, epansion_factor=4, norm_type="IN", activation="gelu", dropout=0.0, spatial_dim=3):
        super().__init__()
        if spatial_dim != 3 or norm_type != "IN" or activation.lower() != "gelu" or not (1 <= len(kernel_sizes) <= 3):
            raise NotImplementedError("veloxseg_amd JLC: 3-D, IN, GELU, 1..3 kernel sizes")
        if len(kernel_sizes) < 2:
            raise NotImplementedError("single-kernel JLC variant (no IN/GELU) is not used by any shipped config")
        self.spatial_convs = nn.ModuleList([
            nn.Sequential(ParamConv3d(in_channels, in_channels, k, padding=k // 2, groups=groups), InstanceNormMarker(in_channels), Marker())
            for k in kernel_sizes])
        self.channel_conv = nn.Sequential(
            InstanceNormMarker(in_channels),
            ParamConv3d(in_channels, in_channels * epansion_factor, 1, 1, 0),
            Marker(),
            ParamConv3d(in_channels * epansion_factor, in_channels, 1, 1, 0),
            nn.Dropout(dropout))
        self.p = dropout
        self.site = VF.new_dropout_site()

    def forward(self, x):
        if VF.USE_COMPOSITE and x.is_cuda:
            return VF.jlc_block(x, self, self.p if self.training else 0.0, self.site)       # same kernels, one autograd node
        ys = [seq[0](x) for seq in self.spatial_convs]
        o = VF.instnorm_sum(ys, act=True, res=x)
        h = VF.gelu_dropout(self.channel_conv[1](VF.instnorm_sum([o])), 0.0, 0)
        return VF.residual_dropout(o, self.channel_conv[3](h), 1.0, self.p if self.training else 0.0, self.site)


def JLCLayer(in_channels, depth=1, kernel_sizes=(1, 3, 5), groups=1, epansion_factor=4, activation="gelu", dropout=0.0, spatial_dim=3):
    return nn.Sequential(*[JLC(in_channels, kernel_sizes=kernel_sizes, groups=groups, epansion_factor=epansion_factor,
                               activation=activation, dropout=dropout, spatial_dim=spatial_dim) for _ in range(depth)])

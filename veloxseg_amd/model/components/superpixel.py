"""PixelShuffle (reference: model/components/superpixel.py:16).  In this build the shuffle is fused into the
store of the producing convolution (vx_conv3d_fwd, ps argument); the module only carries the scale."""
from torch import nn


class PixelShuffle(nn.Module):
    def __init__(self, scale, spatial_dim=3):
        super().__init__()
        if spatial_dim != 3:
            raise NotImplementedError("3-D only (the reference's 2-D pattern is unused and drops the batch axis)")
        self.scale = scale
        self.spatial_dim = spatial_dim

    def forward(self, x):
        raise RuntimeError("PixelShuffle is fused into the preceding convolution; call the parent Sequential's owner")

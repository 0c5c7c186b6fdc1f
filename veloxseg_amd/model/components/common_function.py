"""Small helpers (reference: model/components/common_function.py)."""
from torch import nn

from ... import functional as VF


def get_pram_matrix(x):
    """Gram matrix / (C*H*W*D)  (common_function.py:8-14) -- HIP kernel vx_gram_fwd."""
    if x.dim() != 5:
        raise NotImplementedError("veloxseg_amd implements the 3-D path only")
    return VF.gram(x)


class ParamConv3d(nn.Conv3d):
    """nn.Conv3d used as a PARAMETER HOLDER (same names, shapes, default init and RNG consumption as the
    reference's nn.Conv3d); the arithmetic is the HIP kernel behind veloxseg_amd.functional.conv3d."""

    def forward(self, x, x2=None, pixel_shuffle: int = 1, out_bf16: bool = False):
        return VF.conv3d(x, self.weight, self.bias, x2=x2, stride=self.stride[0], padding=self.padding[0],
                         groups=self.groups, pixel_shuffle=pixel_shuffle, out_bf16=out_bf16)


class ParamConvTranspose3d(nn.ConvTranspose3d):
    def forward(self, x, feeds_instnorm: bool = False):
        assert self.kernel_size == (2, 2, 2) and self.stride == (2, 2, 2) and self.groups == 1
        return VF.conv_transpose_k2s2(x, self.weight, self.bias, feeds_instnorm)


class InstanceNormMarker(nn.Module):
    """Stands where the reference has nn.InstanceNorm3d (no parameters, no buffers): keeps Sequential
    indices -- and therefore state_dict keys -- identical.  The normalisation itself is fused into
    functional.instnorm_sum by the parent module."""

    def __init__(self, channels: int):
        super().__init__()
        self.channels = channels

    def forward(self, x):
        return VF.instnorm_sum([x])


class Marker(nn.Module):
    """Parameter-free placeholder (activation / dropout / pixel-shuffle slots of reference Sequentials)."""

    def forward(self, x):
        raise RuntimeError("placeholder module: the parent block fuses this step")

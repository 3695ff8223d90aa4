"""Parameter shells for the pseudo-3D conv / norm layers (reference guided_diffusion/nn.py:26-87,161-168).
The arithmetic runs in csrc/igemm.hip + csrc/norm.hip; these modules own identically named tensors with the reference's
initialisation (temporal conv = Dirac identity, zero bias)."""
import torch.nn as nn


class GroupNorm32(nn.GroupNorm):
    pass


class Conv3d(nn.Module):
    def __init__(self, dim, dim_out=None, kernel_size=3, stride=(1, 1, 1), *, temporal_kernel_size=None, **kwargs):
        super().__init__()
        dim_out = dim if dim_out is None else dim_out
        temporal_kernel_size = kernel_size if temporal_kernel_size is None else temporal_kernel_size
        if isinstance(stride, int):
            stride = (1, stride, stride)
        self.spatial_conv = nn.Conv2d(dim, dim_out, kernel_size=kernel_size, padding=kernel_size // 2, stride=tuple(stride[1:]))
        self.temporal_conv = nn.Conv1d(dim_out, dim_out, kernel_size=temporal_kernel_size) if kernel_size > 1 else None
        self.kernel_size = kernel_size
        if self.temporal_conv is not None:
            nn.init.dirac_(self.temporal_conv.weight.data)
            nn.init.zeros_(self.temporal_conv.bias.data)


def conv_nd(dims, *args, **kwargs):
    if dims == 1:
        return nn.Conv1d(*args, **kwargs)
    if dims == 2:
        return nn.Conv2d(*args, **kwargs)
    if dims == 3:
        kwargs.pop("padding", None)
        return Conv3d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


def normalization(channels):
    return GroupNorm32(32, channels)

"""Parameter shells (reference model/conv1d_components.py:7-40)."""
import torch.nn as nn


class Downsample1d(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv1d(dim, dim, 3, 2, 1)


class Upsample1d(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.ConvTranspose1d(dim, dim, 4, 2, 1)


class Conv1dBlock(nn.Module):
    """Conv1d --> GroupNorm --> Mish (executed as conv_igemm + fused GroupNorm/Mish/FiLM kernels)."""

    def __init__(self, inp_channels, out_channels, kernel_size, n_groups=8):
        super().__init__()
        self.block = nn.Sequential(nn.Conv1d(inp_channels, out_channels, kernel_size, padding=kernel_size // 2),
                                   nn.GroupNorm(n_groups, out_channels), nn.Mish())

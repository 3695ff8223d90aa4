"""Parameter shells for the pooling nets (reference common/base_nets.py:153-285).  Arithmetic runs in HIP
(csrc/pool.hip: spatial_softmax_fwd/bwd); these modules only own the identically named tensors."""
import numpy as np
import torch
import torch.nn as nn


class Module(nn.Module):
    def output_shape(self, input_shape=None):
        raise NotImplementedError


class ConvBase(Module):
    pass


class SpatialSoftmax(ConvBase):
    def __init__(self, input_shape, num_kp=None, temperature=1.0, learnable_temperature=False, output_variance=False,
                 noise_std=0.0):
        super().__init__()
        assert len(input_shape) == 3
        self._in_c, self._in_h, self._in_w = input_shape
        if learnable_temperature or output_variance or noise_std != 0.0 or num_kp is None or temperature != 1.0:
            raise NotImplementedError("HIP SpatialSoftmax implements the Libero configuration "
                                      "(num_kp set, temperature 1, no variance output, noise_std 0)")
        self.nets = nn.Conv2d(self._in_c, num_kp, kernel_size=1)
        self._num_kp = num_kp
        self.learnable_temperature = False
        self.output_variance = False
        self.noise_std = 0.0
        # the reference registers a (non-trainable) Parameter *as a buffer*; keep the same state_dict entry
        self.register_buffer("temperature", nn.Parameter(torch.ones(1) * temperature, requires_grad=False))
        px, py = np.meshgrid(np.linspace(-1.0, 1.0, self._in_w), np.linspace(-1.0, 1.0, self._in_h))
        self.register_buffer("pos_x", torch.from_numpy(px.reshape(1, self._in_h * self._in_w)).float())
        self.register_buffer("pos_y", torch.from_numpy(py.reshape(1, self._in_h * self._in_w)).float())
        self.kps = None

    def output_shape(self, input_shape):
        return [self._num_kp, 2]

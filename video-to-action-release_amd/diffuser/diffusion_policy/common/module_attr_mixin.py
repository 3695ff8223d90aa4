"""nn.Module base that exposes `.device` / `.dtype` (surface of the reference's common/module_attr_mixin.py:3-14).

The reference answers both from its first registered parameter, an empty placeholder called `_dummy_variable`; the same
placeholder is registered here first (it is part of the checkpoint key layout) and queried directly."""
import torch
from torch import nn


class ModuleAttrMixin(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_parameter("_dummy_variable", nn.Parameter(torch.empty(0)))

    device = property(lambda self: self._dummy_variable.device)
    dtype = property(lambda self: self._dummy_variable.dtype)

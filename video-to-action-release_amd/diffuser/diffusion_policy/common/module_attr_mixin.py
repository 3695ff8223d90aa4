import torch.nn as nn


class ModuleAttrMixin(nn.Module):
    """Same surface as the reference mixin (common/module_attr_mixin.py:3-14): a dummy parameter gives .device/.dtype."""

    def __init__(self):
        super().__init__()
        self._dummy_variable = nn.Parameter()

    @property
    def device(self):
        return next(iter(self.parameters())).device

    @property
    def dtype(self):
        return next(iter(self.parameters())).dtype

from typing import Dict
import torch
from .common.module_attr_mixin import ModuleAttrMixin


class BaseImagePolicy(ModuleAttrMixin):
    def predict_action(self, obs_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        raise NotImplementedError()

    def reset(self):
        pass

    def set_normalizer(self, normalizer):
        raise NotImplementedError()

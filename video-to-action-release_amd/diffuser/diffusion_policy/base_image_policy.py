"""Abstract image-policy surface (reference base_image_policy.py:7-26): predict_action / reset / set_normalizer."""
from .common.module_attr_mixin import ModuleAttrMixin


class BaseImagePolicy(ModuleAttrMixin):
    def predict_action(self, obs_dict):
        """obs_dict: {key: [B, To, ...]} -> {'action': [B, Ta, Da], ...}"""
        raise NotImplementedError

    def reset(self):
        """stateless policies: nothing to do"""

    def set_normalizer(self, normalizer):
        raise NotImplementedError

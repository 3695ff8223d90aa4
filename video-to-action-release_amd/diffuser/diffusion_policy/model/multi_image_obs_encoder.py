"""Parameter shell of MultiImageObsEncoder (reference model/multi_image_obs_encoder.py:11-212): one independent
VisualCore per rgb key (share_rgb_model False), BatchNorm2d swapped for GroupNorm(C//16, C) (:66-74), keys iterated
in sorted order (:132).  forward() runs the HIP encoder through the owning policy's engine."""
import copy
import torch
import torch.nn as nn
from ..common.module_attr_mixin import ModuleAttrMixin


def _swap_bn_for_gn(root: nn.Module):
    for name, child in list(root.named_children()):
        if isinstance(child, nn.BatchNorm2d):
            setattr(root, name, nn.GroupNorm(num_groups=child.num_features // 16, num_channels=child.num_features))
        else:
            _swap_bn_for_gn(child)
    return root


class MultiImageObsEncoder(ModuleAttrMixin):
    def __init__(self, shape_meta, rgb_model, resize_shape=None, crop_shape=None, random_crop=True, use_group_norm=False,
                 share_rgb_model=False, imagenet_norm=False, _target_=None):
        super().__init__()
        if share_rgb_model or not use_group_norm or resize_shape is not None or crop_shape is not None or imagenet_norm:
            raise NotImplementedError("only the Libero configuration (independent GroupNorm encoders, no resize/crop) is on the path")
        key_model_map = nn.ModuleDict()
        key_shape_map = {}
        rgb_keys = []
        for key, attr in shape_meta["obs"].items():
            key_shape_map[key] = tuple(attr["shape"])
            if attr.get("type", "low_dim") != "rgb":
                raise NotImplementedError("low_dim / mlp observations are not on the Libero path")
            rgb_keys.append(key)
            model = rgb_model[key] if isinstance(rgb_model, dict) else copy.deepcopy(rgb_model)
            key_model_map[key] = _swap_bn_for_gn(model)
        self.shape_meta = shape_meta
        self.key_model_map = key_model_map
        self.share_rgb_model = False
        self.rgb_keys = sorted(rgb_keys)
        self.low_dim_keys = []
        self.key_shape_map = key_shape_map

    def output_shape(self):
        dims = 0
        for key in self.rgb_keys:
            dims += self.key_model_map[key].output_shape(self.key_shape_map[key])[0]
        return (dims,)

    @torch.no_grad()
    def forward(self, obs_dict):
        """obs_dict[key]: [B,3,H,W] ALREADY normalised to [-1,1] (as the reference feeds it) -> [B, sum(feature)]."""
        owner = object.__getattribute__(self, "__dict__").get("_owner_ref")
        if owner is None or owner() is None:
            raise RuntimeError("MultiImageObsEncoder.forward needs its owning DiffusionUnetImagePolicy (HIP engine)")
        return owner()._encode_normalized(obs_dict)

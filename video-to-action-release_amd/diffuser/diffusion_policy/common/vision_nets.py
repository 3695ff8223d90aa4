"""Parameter shells of the image encoder (reference common/vision_nets.py:9-177).  ResNet-18 topology follows the
published torchvision 0.15.1 architecture (third-party, absent from the reference tree); parameter names and child
order are the compatibility contract: backbone.nets.{0,1,4..7}..., pool.nets, nets.3."""
import math
import numpy as np
import torch
import torch.nn as nn
from .base_nets import ConvBase, SpatialSoftmax, Module  # noqa: F401


class _BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))


def _resnet18_children(input_channel=3, widths=(64, 128, 256, 512)):
    """children()[:-2] of torchvision resnet18: conv1, bn1, relu, maxpool, layer1..4 (kaiming fan_out init)."""
    mods = [nn.Conv2d(input_channel, widths[0], 7, 2, 3, bias=False), nn.BatchNorm2d(widths[0]), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, 2, 1)]
    inpl = widths[0]
    for li, c in enumerate(widths):
        mods.append(nn.Sequential(_BasicBlock(inpl, c, 1 if li == 0 else 2), _BasicBlock(c, c, 1)))
        inpl = c
    seq = nn.Sequential(*mods)
    for m in seq.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return seq


class ResNet18Conv(ConvBase):
    def __init__(self, input_channel=3, pretrained=False, input_coord_conv=False, widths=(64, 128, 256, 512)):
        super().__init__()
        if input_coord_conv or pretrained:
            raise NotImplementedError("coord-conv / pretrained weights are not on the Libero path")
        self._input_channel = input_channel
        self._widths = tuple(widths)
        self.nets = _resnet18_children(input_channel, widths)

    def output_shape(self, input_shape):
        return [self._widths[-1], int(math.ceil(input_shape[1] / 32.0)), int(math.ceil(input_shape[2] / 32.0))]


class VisualCore(ConvBase):
    def __init__(self, input_shape, backbone_class, backbone_kwargs, pool_class=None, pool_kwargs=None, flatten=True,
                 feature_dimension=None, **kwargs):
        super().__init__()
        assert backbone_class == "ResNet18Conv" and pool_class == "SpatialSoftmax" and flatten
        self.input_shape = input_shape
        self.flatten = flatten
        bk = dict(backbone_kwargs or {})
        self.backbone = ResNet18Conv(input_channel=input_shape[0], pretrained=bool(bk.get("pretrained")),
                                     input_coord_conv=bool(bk.get("input_coord_conv")),
                                     widths=bk.get("widths", (64, 128, 256, 512)))
        feat_shape = self.backbone.output_shape(input_shape)
        pk = dict(pool_kwargs or {})
        pk.pop("input_shape", None)
        self.pool = SpatialSoftmax(input_shape=feat_shape, **pk)
        feat_shape = self.pool.output_shape(feat_shape)
        net_list = [self.backbone, self.pool, nn.Flatten(start_dim=1, end_dim=-1)]
        self.feature_dimension = feature_dimension
        if feature_dimension is not None:
            net_list.append(nn.Linear(int(np.prod(feat_shape)), feature_dimension))
        self.nets = nn.Sequential(*net_list)

    def output_shape(self, input_shape):
        if self.feature_dimension is not None:
            return [self.feature_dimension]
        return [int(np.prod(self.pool.output_shape(self.backbone.output_shape(input_shape))))]

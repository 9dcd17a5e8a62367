"""ResNet trunks with torchvision's parameter names, without torchvision.

Mirror of the reference's salve/models/resnet_factory.py:7-51.  The reference builds its trunk with
`torchvision.models.resnet{18,34,50,152}`; torchvision is not a dependency here, so the module tree is restated
with exactly torchvision's attribute names (conv1, bn1, layer1..4.<i>.conv{1,2,3}, bn{1,2,3}, downsample.{0,1},
fc) -- that keeps the reference's checkpoints (`state_dict` keys under `resnet.`) loadable with strict=True.
These modules only HOLD parameters; the forward pass runs in salve_amd/csrc/resnet.hip.
"""

from __future__ import annotations

from typing import List, Tuple

import torch
from torch import nn

# (block type, blocks per stage) -- torchvision ResNet v1.5 definitions
RESNET_SPECS = {
    18: ("basic", [2, 2, 2, 2]),
    34: ("basic", [3, 4, 6, 3]),
    50: ("bottleneck", [3, 4, 6, 3]),
    101: ("bottleneck", [3, 4, 23, 3]),
    152: ("bottleneck", [3, 8, 36, 3]),
}


def get_resnet_feature_dim(num_layers: int) -> int:
    """Width of the pooled feature vector (reference resnet_factory.py:7-23)."""
    if num_layers in [18, 34]:
        return 512
    if num_layers in [50, 101, 152]:
        return 2048
    raise RuntimeError("Num layers not allowed")


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int, downsample: bool) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)  # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.stride = stride
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))


class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int, downsample: bool) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.stride = stride
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes))


class ResNetTrunk(nn.Module):
    """Parameter container equal in layout to torchvision.models.ResNet (1000-way fc included, as the reference
    keeps it in its checkpoints although forward never uses it, early_fusion.py:20,67,81)."""

    def __init__(self, num_layers: int) -> None:
        super().__init__()
        kind, blocks = RESNET_SPECS[num_layers]
        Block = _Bottleneck if kind == "bottleneck" else _BasicBlock
        self.block_kind = kind
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = 64
        for si, (planes, n) in enumerate(zip([64, 128, 256, 512], blocks)):
            layers: List[nn.Module] = []
            for bi in range(n):
                stride = 2 if (bi == 0 and si > 0) else 1
                need_ds = bi == 0 and (stride != 1 or inplanes != planes * Block.expansion)
                layers.append(Block(inplanes, planes, stride, need_ds))
                inplanes = planes * Block.expansion
            setattr(self, f"layer{si + 1}", nn.Sequential(*layers))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(inplanes, 1000)
        # torchvision's initialisation
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


def get_vanilla_resnet_model(num_layers: int, pretrained: bool) -> nn.Module:
    """Reference resnet_factory.py:26-48.  ImageNet weights cannot be downloaded here (no torchvision, no
    network): with pretrained=True the trunk is still randomly initialised and a warning is issued -- every
    released SALVe checkpoint overwrites these weights anyway (train_utils.py:229-242)."""
    assert num_layers in [18, 34, 50, 101, 152]
    if num_layers == 101:
        raise RuntimeError("num layers not supported")  # the reference factory has no 101 branch either (:37-46)
    if pretrained:
        import warnings

        warnings.warn("ImageNet-pretrained weights are unavailable offline; the ResNet trunk is randomly initialised")
    return ResNetTrunk(num_layers)

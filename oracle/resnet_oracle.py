"""CPU restatement (torch fp32) of the verifier -- TEST INFRASTRUCTURE, see oracle/__init__.py.

Reference: salve/models/early_fusion.py:41-83 (forward) on top of the torchvision ResNet selected by
salve/models/resnet_factory.py:26-44.  torchvision is an un-vendored, unpinned dependency of the reference and is not
installed here ("parity unpinned"); its published ResNet v1.5 definition is restated functionally:
Bottleneck (expansion 4, stride on the 3x3 conv) / BasicBlock, blocks [3,4,6,3] (50), [3,8,36,3] (152), [2,2,2,2] (18),
[3,4,6,3] (34); downsample = 1x1 conv(stride) + BN on the first block of a stage; BN eps 1e-5 (eval mode: running
statistics); maxpool 3x3/2 pad 1; AdaptiveAvgPool(1); the 1000-way resnet.fc and resnet.conv1 are present in the
checkpoint but bypassed (early_fusion.py:20,67,81).
Cross-check (round 5): HuggingFace `transformers.ResNetModel` -- an independent implementation of the same published network, in the
image -- loaded with the same tensors agrees with `forward` to float32 rounding for ResNet-18 / 34 / 50 / 152
(tests/test_oracle_hf_resnet.py).  That pins the STRUCTURE against a second implementation; it is still not torchvision itself.
"""

from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

BLOCKS = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]),
          152: ("bottleneck", [3, 8, 36, 3])}


def _bn(x, sd, p):
    return F.batch_norm(x, sd[f"{p}.running_mean"], sd[f"{p}.running_var"], sd[f"{p}.weight"], sd[f"{p}.bias"], False, 0.0, 1e-5)


def forward(sd: Dict[str, torch.Tensor], num_layers: int, xs: List[torch.Tensor]) -> torch.Tensor:
    """xs: the 2 / 4 / 6 normalised [B,3,224,224] fp32 tiles.  Returns fp32 logits [B, num_classes]."""
    sd = {(k[7:] if k.startswith("module.") else k): v.float() for k, v in sd.items()}
    kind, blocks = BLOCKS[num_layers]
    x = torch.cat(xs, dim=1)  # early_fusion.py:55-65
    x = F.conv2d(x, sd["conv1.weight"], None, stride=2, padding=3)  # :67
    x = F.relu(_bn(x, sd, "resnet.bn1"))  # :69-70
    x = F.max_pool2d(x, 3, 2, 1)  # :71
    for si, n in enumerate(blocks):  # :73-76
        for bi in range(n):
            p = f"resnet.layer{si + 1}.{bi}"
            stride = 2 if (bi == 0 and si > 0) else 1
            idn = x
            if kind == "bottleneck":
                o = F.relu(_bn(F.conv2d(x, sd[f"{p}.conv1.weight"]), sd, f"{p}.bn1"))
                o = F.relu(_bn(F.conv2d(o, sd[f"{p}.conv2.weight"], None, stride, 1), sd, f"{p}.bn2"))
                o = _bn(F.conv2d(o, sd[f"{p}.conv3.weight"]), sd, f"{p}.bn3")
            else:
                o = F.relu(_bn(F.conv2d(x, sd[f"{p}.conv1.weight"], None, stride, 1), sd, f"{p}.bn1"))
                o = _bn(F.conv2d(o, sd[f"{p}.conv2.weight"], None, 1, 1), sd, f"{p}.bn2")
            if f"{p}.downsample.0.weight" in sd:
                idn = _bn(F.conv2d(x, sd[f"{p}.downsample.0.weight"], None, stride), sd, f"{p}.downsample.1")
            x = F.relu(o + idn)
    x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)  # :78-79
    return F.linear(x, sd["fc.weight"], sd["fc.bias"])  # :81


def expected_state_dict_keys(num_layers: int, num_images: int) -> List[str]:
    """Key list the reference's checkpoints carry, used to test the product module's layout."""
    kind, blocks = BLOCKS[num_layers]
    keys = ["conv1.weight", "fc.weight", "fc.bias", "resnet.conv1.weight"]
    bn = lambda p: [f"{p}.{s}" for s in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
    keys += bn("resnet.bn1")
    nconv = 3 if kind == "bottleneck" else 2
    for si, n in enumerate(blocks):
        for bi in range(n):
            p = f"resnet.layer{si + 1}.{bi}"
            for c in range(1, nconv + 1):
                keys += [f"{p}.conv{c}.weight"] + bn(f"{p}.bn{c}")
            if bi == 0 and (si > 0 or kind == "bottleneck"):
                keys += [f"{p}.downsample.0.weight"] + bn(f"{p}.downsample.1")
    keys += ["resnet.fc.weight", "resnet.fc.bias"]
    return keys

"""Early-fusion ResNet verifier with the reference's module interface, executed by HIP kernels.

Mirror of salve/models/early_fusion.py:11-83.  Same constructor, same `forward(x1..x6)`, same parameter
names (`conv1.weight`, `fc.{weight,bias}`, `resnet.*`) so the reference's checkpoints load with strict=True
(train_utils.py:229-242).  `forward` is inference-only: it folds BatchNorm with the running statistics
(eval-mode semantics, what scripts/test.py runs under torch.no_grad) and executes the network in
salve_amd/csrc/resnet.hip.  There is no CPU or eager fallback: without the HIP library, or on a CPU tensor,
`forward` raises.
"""

from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

import salve_amd.models.resnet_factory as resnet_factory
from salve_amd.models.hip_resnet import HipResNet, nchw_to_input

_TWO = [{"layout"}, {"ceiling_rgb_texture"}, {"floor_rgb_texture"}]
_FOUR = {"ceiling_rgb_texture", "floor_rgb_texture"}
_SIX = {"ceiling_rgb_texture", "floor_rgb_texture", "layout"}


def num_input_images(modalities) -> int:
    m = set(modalities)
    if m in _TWO:
        return 2
    if m == _FOUR:
        return 4
    if m == _SIX:
        return 6
    raise RuntimeError(f"Unsupported modalities. {str(modalities)}")


class EarlyFusionCEResnet(nn.Module):
    """Early-fusion model designed for a cross-entropy (CE) loss."""

    def __init__(self, num_layers: int, pretrained: bool, num_classes: int, args) -> None:
        super().__init__()
        assert num_classes > 1
        self.modalities = args.modalities
        self.num_layers = num_layers
        self.resnet = resnet_factory.get_vanilla_resnet_model(num_layers, pretrained)
        self.inplanes = 64
        self.num_images = num_input_images(self.modalities)
        self.conv1 = nn.Conv2d(3 * self.num_images, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.fc = nn.Linear(resnet_factory.get_resnet_feature_dim(num_layers), num_classes)
        self._compiled: Optional[HipResNet] = None
        self._compiled_key = None

    # ------------------------------------------------------------------ HIP engine
    def _state_key(self, device: torch.device):
        return (str(device), tuple((k, v._version, v.data_ptr()) for k, v in self.state_dict().items()))

    def compiled(self, device: torch.device, flags: int = 0) -> HipResNet:
        """Fold + pack the current weights for `device` (cached until a parameter changes).  flags: the library's kernel
        selection (_lib.RESNET_*; 0 = the product's -- development tools and the bit-identity tests pass others)."""
        key = (self._state_key(device), int(flags))
        if self._compiled is None or self._compiled_key != key:
            self._compiled = HipResNet(self.state_dict(), self.num_layers, device, flags=flags)
            self._compiled_key = key
        return self._compiled

    def forward(self, x1: Tensor, x2: Tensor, x3: Optional[Tensor], x4: Optional[Tensor], x5: Optional[Tensor],
                x6: Optional[Tensor]) -> torch.Tensor:
        """Early fusion = concatenation along channels (early_fusion.py:55-65), then the ResNet.
        Inputs must be finite: the HIP kernels' ReLUs turn a NaN activation into 0 instead of propagating it to the logits
        as torch does (salve_amd/csrc/resnet.hip: track4).  Magnitudes beyond fp16 are reported through `check()`."""
        n = num_input_images(self.modalities)  # raises RuntimeError on unsupported sets, like the reference
        xs = [x1, x2, x3, x4, x5, x6][:n]
        if any(x is None for x in xs):
            raise RuntimeError(f"{n} input images are required for modalities {self.modalities}")
        if x1.device.type != "cuda":
            raise RuntimeError("EarlyFusionCEResnet.forward runs on the HIP device only (no CPU fallback)")
        if self.training and torch.is_grad_enabled():
            raise RuntimeError("the HIP verifier is inference-only: call model.eval() / use torch.no_grad()")
        eng = self.compiled(x1.device)
        return eng.forward_nhwc(nchw_to_input(xs, eng.in_channels))

    def check(self, device=None, what: str = "EarlyFusionCEResnet.forward") -> None:
        """The forward is asynchronous: an activation beyond the fp16 range is saturated AND reported in the device status word
        (salve_hip.h).  Call this where the host reads the logits (it synchronises); it raises instead of letting the logits of
        a different network through.  evaluate.run_test_epoch / run_fused_epoch and the pipeline do."""
        from salve_amd import status

        status.check(device if device is not None else torch.device("cuda", torch.cuda.current_device()), what)

    def forward_nhwc(self, x: Tensor) -> Tensor:
        """Fused-pipeline entry: fp16 [B,224,224,Cpad] tiles written by the rasteriser -> fp32 logits."""
        return self.compiled(x.device).forward_nhwc(x)

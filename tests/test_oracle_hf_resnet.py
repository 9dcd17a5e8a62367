"""The ResNet restatement of oracle/resnet_oracle.py against an INDEPENDENT implementation of the published architecture.

torchvision -- the reference's own dependency (resnet_factory.py:26-44) -- is not in the image, so the oracle's verifier leg cannot be
pinned by the reference itself.  HuggingFace `transformers` is, and its `ResNetModel` (the v1.5 layout with
`downsample_in_bottleneck=False`: stride on the 3x3 convolution; the model its `microsoft/resnet-*` checkpoints were converted into from
torchvision / timm weights) is a second, unrelated implementation of the same published network.  The reference's checkpoint keys are
mapped onto it tensor for tensor and the two forward passes must agree to float32 rounding -- for BasicBlock (18, 34) and Bottleneck
(50, 152) networks, 6 / 12 input channels, trained-looking BatchNorm statistics.  What this pins: block structure, strides, padding,
where the ReLUs sit, the shortcut, eps, the pooling -- the things a restatement can get wrong.  CPU only."""
from types import SimpleNamespace

import pytest
import torch

from oracle import resnet_oracle as ro
from salve_amd.models.early_fusion import EarlyFusionCEResnet
from tests._helpers import randomise_bn

transformers = pytest.importorskip("transformers")

MODS = {2: ["floor_rgb_texture"], 4: ["ceiling_rgb_texture", "floor_rgb_texture"]}


def hf_from_reference_checkpoint(sd, num_layers: int, in_channels: int):
    """transformers.ResNetModel holding the tensors of a reference-layout state dict (conv1.*, resnet.layerS.B.*)."""
    kind, blocks = ro.BLOCKS[num_layers]
    widths = [256, 512, 1024, 2048] if kind == "bottleneck" else [64, 128, 256, 512]
    cfg = transformers.ResNetConfig(num_channels=in_channels, embedding_size=64, hidden_sizes=widths, depths=list(blocks), layer_type=kind,
                                    hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    model = transformers.ResNetModel(cfg).eval()
    bn = lambda src, dst: {f"{dst}.{s}": sd[f"{src}.{s}"] for s in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")}
    new = {"embedder.embedder.convolution.weight": sd["conv1.weight"], **bn("resnet.bn1", "embedder.embedder.normalization")}
    for si, n in enumerate(blocks):
        for bi in range(n):
            p, q = f"resnet.layer{si + 1}.{bi}", f"encoder.stages.{si}.layers.{bi}"
            for k in range(3 if kind == "bottleneck" else 2):
                new[f"{q}.layer.{k}.convolution.weight"] = sd[f"{p}.conv{k + 1}.weight"]
                new.update(bn(f"{p}.bn{k + 1}", f"{q}.layer.{k}.normalization"))
            if f"{p}.downsample.0.weight" in sd:
                new[f"{q}.shortcut.convolution.weight"] = sd[f"{p}.downsample.0.weight"]
                new.update(bn(f"{p}.downsample.1", f"{q}.shortcut.normalization"))
    missing, unexpected = model.load_state_dict(new, strict=True), None
    assert not missing.missing_keys and not missing.unexpected_keys
    return model


@pytest.mark.parametrize("num_layers,n_images,hw", [(18, 2, 224), (34, 2, 96), (50, 2, 224), (50, 4, 96), (152, 4, 96)])
def test_oracle_resnet_equals_the_transformers_implementation(num_layers, n_images, hw):
    torch.manual_seed(3)
    model = EarlyFusionCEResnet(num_layers, False, 2, SimpleNamespace(modalities=MODS[n_images]))
    randomise_bn(model, seed=num_layers)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    xs = [torch.randn(2, 3, hw, hw) for _ in range(n_images)]
    with torch.no_grad():
        want = ro.forward(sd, num_layers, xs)
        hf = hf_from_reference_checkpoint(sd, num_layers, 3 * n_images)
        feat = hf(torch.cat(xs, 1)).pooler_output.flatten(1)
        got = torch.nn.functional.linear(feat, sd["fc.weight"], sd["fc.bias"])
    assert got.shape == want.shape == (2, 2)
    scale = float(want.abs().max())
    assert scale > 1e-3
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, scale), (got, want)

"""Logit error of the HIP verifier against the fp32 oracle fed the SAME fp32 tiles (no quantisation on the oracle side),
for trained-looking and for default BatchNorm statistics (GPU box).  Prints one line per case."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from types import SimpleNamespace
import torch
from oracle import resnet_oracle as ro
from salve_amd.models.early_fusion import EarlyFusionCEResnet
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from _helpers import randomise_bn

DEV = "cuda:0"
for layers, mods, batch in ((18, ["layout"], 3), (50, ["floor_rgb_texture"], 4), (152, ["ceiling_rgb_texture", "floor_rgb_texture"], 2)):
    for bn in ("random", "default"):
        torch.manual_seed(0)
        model = EarlyFusionCEResnet(layers, False, 2, SimpleNamespace(modalities=mods))
        if bn == "random":
            randomise_bn(model)
        model.eval()
        n = len(mods) * 2
        # tile-like inputs: normalised uint8 values
        v = torch.randint(0, 256, (n, batch, 3, 224, 224)).float()
        mean = torch.tensor([123.675, 116.28, 103.53]).view(1, 1, 3, 1, 1)
        std = torch.tensor([58.395, 57.12, 57.375]).view(1, 1, 3, 1, 1)
        xs = list(((v - mean) / std).unbind(0))
        with torch.no_grad():
            ref32 = ro.forward(model.state_dict(), layers, xs)
            ref16 = ro.forward(model.state_dict(), layers, [x.half().float() for x in xs])
            pad = xs + [None] * (6 - n)
            got = model.to(DEV)(*[None if x is None else x.to(DEV) for x in pad]).cpu()
        print(f"resnet{layers} bn={bn}: |logit|max {float(ref32.abs().max()):.3f}  err vs fp32-input oracle {float((got - ref32).abs().max()):.2e}"
              f"  vs fp16-input oracle {float((got - ref16).abs().max()):.2e}  oracle fp32-vs-fp16 input {float((ref32 - ref16).abs().max()):.2e}", flush=True)

import os
"""Verifier-only timing at several batch sizes (GPU box)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from types import SimpleNamespace
import torch
from salve_amd.models.early_fusion import EarlyFusionCEResnet
dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 50
mods = ["floor_rgb_texture"] if layers != 152 else ["ceiling_rgb_texture", "floor_rgb_texture"]
model = EarlyFusionCEResnet(layers, False, 2, SimpleNamespace(modalities=mods)).eval()
eng = model.compiled(dev, flags=int(os.environ.get("SALVE_RESNET_FLAGS", "0")))
flop = {50: 8.41e9, 152: 23.73e9, 18: 3.6e9}[layers]
for B in ((32, 64, 128, 256, 512) if len(sys.argv) < 3 else tuple(int(v) for v in sys.argv[2].split(","))):
    x = torch.randn(B, 224, 224, eng.in_channels, device=dev).to(torch.float16)
    for _ in range(2):
        eng.forward_nhwc(x)
    torch.cuda.synchronize()
    reps = max(2, 2048 // B)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.forward_nhwc(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"resnet{layers} B={B}: {dt*1e3:.2f} ms/forward, {B/dt:.0f} samples/s, {B/dt*flop/1e12:.0f} TFLOP/s")

"""One ResNet-50 forward at B=512 after warm-up (GPU box; run under rocprofv3 --kernel-trace to get per-launch times)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from types import SimpleNamespace
import torch
from salve_amd.models.early_fusion import EarlyFusionCEResnet
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
eng = model.compiled(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.randn(B, 224, 224, eng.in_channels, device=dev).to(torch.float16)
for _ in range(3):
    eng.forward_nhwc(x)
torch.cuda.synchronize()

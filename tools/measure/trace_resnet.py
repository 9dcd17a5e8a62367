import os
"""One ResNet-50 forward after warm-up (GPU box; run under rocprofv3 --kernel-trace / --pmc to get per-launch numbers).
usage: trace_resnet.py [batch] [ops.json]   -- ops.json receives the op program (shapes) the launches execute."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from types import SimpleNamespace
import torch
from salve_amd.models.early_fusion import EarlyFusionCEResnet
from salve_amd.models import hip_resnet
dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 50
mods = ["floor_rgb_texture"] if layers != 152 else ["ceiling_rgb_texture", "floor_rgb_texture"]
model = EarlyFusionCEResnet(layers, False, 2, SimpleNamespace(modalities=mods)).eval()
hip_resnet.CHUNK_MAJOR_K = os.environ.get("SALVE_K_ORDER", "") == "chunk"
eng = model.compiled(dev, flags=int(os.environ.get("SALVE_RESNET_FLAGS", "0")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
if len(sys.argv) > 2:
    ops = hip_resnet.build_program(model.state_dict(), layers)[0]
    Path(sys.argv[2]).write_text(json.dumps([{k: int(o[k]) for k in ops.dtype.names} for o in ops]))
x = torch.randn(B, 224, 224, eng.in_channels, device=dev).to(torch.float16)
for _ in range(3):
    eng.forward_nhwc(x)
torch.cuda.synchronize()

"""Per-launch time of the expand_chain kernels in one forward at batch 4096 under development switches (GPU box).
usage: bench_chain.py  (SALVE_RESNET_FLAGS: 64 = no chain, 128 = expand only, 256 = 16 waves, 512 = unsplit; SALVE_CHAIN_DBG works in an ablation build
loaded with SALVE_HIP_LIB only)"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from types import SimpleNamespace
import torch
from salve_amd.models.early_fusion import EarlyFusionCEResnet
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
eng = model.compiled(dev, flags=int(os.environ.get("SALVE_RESNET_FLAGS", "0")))
B = 4096
x = torch.randn(B, 224, 224, eng.in_channels, device=dev).to(torch.float16)
for _ in range(2):
    eng.forward_nhwc(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng.forward_nhwc(x)
torch.cuda.synchronize()
print(f"dbg={os.environ.get('SALVE_CHAIN_DBG', '0')} flags={os.environ.get('SALVE_RESNET_FLAGS', '0')}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms/forward")

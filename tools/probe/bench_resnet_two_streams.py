import os
"""Experiment (GPU box): two verifier forwards on two HIP streams, the second one started half a forward later, so that the
HBM-bound layers of one overlap the MFMA-bound layers of the other.  Compared with the same work on one stream."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from types import SimpleNamespace
import torch
from salve_amd.models.early_fusion import EarlyFusionCEResnet
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
e1, e2 = model.compiled(dev, flags=int(os.environ.get("SALVE_RESNET_FLAGS", "0"))), model.compiled(dev, flags=int(os.environ.get("SALVE_RESNET_FLAGS", "0")))
x1 = torch.randn(B, 224, 224, e1.in_channels, device=dev).to(torch.float16)
x2 = torch.randn(B, 224, 224, e1.in_channels, device=dev).to(torch.float16)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for _ in range(2):
    e1.forward_nhwc(x1); e2.forward_nhwc(x2)
torch.cuda.synchronize()
R = 6
t0 = time.perf_counter()
for _ in range(R):
    e1.forward_nhwc(x1); e2.forward_nhwc(x2)
torch.cuda.synchronize()
one = (time.perf_counter() - t0) / R
print(f"one stream:  2 x {B}: {one*1e3:.2f} ms  ({2*B/one:.0f} samples/s)")
half = torch.empty(B // 2, 224, 224, e1.in_channels, device=dev, dtype=torch.float16).normal_()
for offset in (False, True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if offset:
        with torch.cuda.stream(s2):
            e2.forward_nhwc(half)          # puts stream 2 half a forward behind (extra work, not counted below)
    for _ in range(R):
        with torch.cuda.stream(s1):
            e1.forward_nhwc(x1)
        with torch.cuda.stream(s2):
            e2.forward_nhwc(x2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    extra = one * 0.25 if offset else 0.0    # the lead-in forward of B/2 samples ~ a quarter of the pair
    print(f"two streams{' (offset)' if offset else ''}: {R} x 2 x {B} in {dt*1e3:.1f} ms -> {(dt-extra)/R*1e3:.2f} ms per pair")

"""Development: the verifier forward with the 3 x 3 convolutions' K ordered (tap, channel) -- the product's -- and (64-channel chunk,
tap, channel in chunk), alternating in one process; logits compared (same sums in another order: close, not equal).
usage: python tools/probe/ab_korder.py <layers> <batch> [flags]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from types import SimpleNamespace
import torch
from salve_amd.models.early_fusion import EarlyFusionCEResnet
from salve_amd.models import hip_resnet
dev = torch.device("cuda:0")
torch.manual_seed(0)
layers, B = int(sys.argv[1]), int(sys.argv[2])
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mods = ["floor_rgb_texture"] if layers != 152 else ["ceiling_rgb_texture", "floor_rgb_texture"]
model = EarlyFusionCEResnet(layers, False, 2, SimpleNamespace(modalities=mods)).eval()
engs = {}
for order in (False, True):
    hip_resnet.CHUNK_MAJOR_K = order
    engs[order] = hip_resnet.HipResNet(model.state_dict(), layers, dev, flags=flags)
x = torch.randn(B, 224, 224, engs[False].in_channels, device=dev).to(torch.float16)
outs = {}
for o, e in engs.items():
    for _ in range(2):
        outs[o] = e.forward_nhwc(x).clone()
    torch.cuda.synchronize()
print("max |logit difference| between the two orders:", float((outs[True] - outs[False]).abs().max()), "max |logit|", float(outs[False].abs().max()))
times = {o: [] for o in engs}
for rnd in range(6):
    for o, e in engs.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            e.forward_nhwc(x)
        torch.cuda.synchronize()
        times[o].append((time.perf_counter() - t0) / 4 * 1e3)
for o in engs:
    t = times[o]
    print(f"resnet{layers} B={B} chunk-major K={o}: " + " ".join(f"{v:.2f}" for v in t) + f"  | mean of last 5 {sum(t[1:])/5:.2f} ms, min {min(t):.2f} ms", flush=True)

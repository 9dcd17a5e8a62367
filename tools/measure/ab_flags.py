"""Development: the verifier forward under several `flags` values of salve_resnet_create, alternating in ONE process on one box
(box-to-box and run-to-run drift is larger than most of the differences looked for).
usage: python tools/measure/ab_flags.py <layers> <batch> <flags> [<flags> ...]      e.g.  ab_flags.py 50 4096 0 1024"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from types import SimpleNamespace
import torch
from salve_amd.models.early_fusion import EarlyFusionCEResnet
dev = torch.device("cuda:0")
torch.manual_seed(0)
layers, B = int(sys.argv[1]), int(sys.argv[2])
flag_list = [int(v) for v in sys.argv[3:]]
mods = ["floor_rgb_texture"] if layers != 152 else ["ceiling_rgb_texture", "floor_rgb_texture"]
model = EarlyFusionCEResnet(layers, False, 2, SimpleNamespace(modalities=mods)).eval()
engs = {f: model.compiled(dev, flags=f) for f in flag_list}
x = torch.randn(B, 224, 224, engs[flag_list[0]].in_channels, device=dev).to(torch.float16)
ref = None
for f, e in engs.items():
    for _ in range(2):
        o = e.forward_nhwc(x).clone()
    torch.cuda.synchronize()
    assert ref is None or torch.equal(o, ref), f"flags {f}: logits differ"
    ref = o
times = {f: [] for f in flag_list}
for rnd in range(6):
    for f, e in engs.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            e.forward_nhwc(x)
        torch.cuda.synchronize()
        times[f].append((time.perf_counter() - t0) / 4 * 1e3)
for f in flag_list:
    t = times[f]
    print(f"resnet{layers} B={B} flags={f}: " + " ".join(f"{v:.2f}" for v in t) + f"  | mean of last 5 {sum(t[1:])/5:.2f} ms, min {min(t):.2f} ms", flush=True)

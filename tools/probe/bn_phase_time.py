"""Where a tile of bottleneck_kernel spends its cycles (GPU box).  Needs a library built with -DSALVE_BN_TIMERS
(tools/probe/build_bn_timers.sh -> tools/probe/_abl/libsalve_bn_timers.so; pass it as SALVE_HIP_LIB): wave 0 of every workgroup sums s_memtime
laps per phase, read back through salve_debug_bn_timers."""
import ctypes, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from types import SimpleNamespace
import torch
from salve_amd import _lib
from salve_amd.models.early_fusion import EarlyFusionCEResnet

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
eng = model.compiled(dev, flags=int(os.environ.get("SALVE_RESNET_FLAGS", "0")))
x = torch.randn(B, 224, 224, eng.in_channels, device=dev).to(torch.float16)
lib = _lib.load()
fn = lib.salve_debug_bn_timers
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
out = (ctypes.c_ulonglong * 32)()
eng.forward_nhwc(x); torch.cuda.synchronize(); fn(out)
eng.forward_nhwc(x); torch.cuda.synchronize(); fn(out)
names = ["first X stage wait", "rest of GEMM 1", "t1 epilogue", "first Wb stage wait", "rest of GEMM 2", "t2 epilogue", "GEMM 3 + stores"]
for form, label in ((1, "PROJ form (1 launch)"), (0, "plain form (2 launches)")):
    v = [out[form * 16 + k] for k in range(16)]
    tiles, tot = max(1, v[7]), sum(v[:7])
    print(f"{label}: {tiles} tiles, {tot / tiles:.0f} s_memtime ticks per tile")
    for k in range(7):
        print(f"  {names[k]:22s} {v[k] / tiles:8.0f}  {100.0 * v[k] / tot:5.1f} %")
    for k, name in ((8, "weight chunk wait"), (9, "MFMAs + epilogue"), (10, "barrier, stores, barrier")):
        print(f"    GEMM 3: {name:24s} {v[k] / tiles:8.0f}  {100.0 * v[k] / tot:5.1f} %")

"""GPU box: HBM copy rate of the float4 copy kernel (tests/native) over its launch parameters, beside torch's copy_ (the runtime's blit).
usage: python tools/measure/copy_sweep.py [GB = 4]"""
import ctypes, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
lib = ctypes.CDLL(str(ROOT / "tests" / "native" / "libsalve_testhelp.so"))
lib.salve_debug_copy16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
dev = torch.device("cuda:0")
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gb * (1 << 30)) // 16
src = torch.empty((n, 4), dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
cus = torch.cuda.get_device_properties(dev).multi_processor_count
def timed(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return 2.0 * n * 16 / (e0.elapsed_time(e1) / iters * 1e-3) / 1e9
print(f"torch copy_: {timed(lambda: dst.copy_(src)):.0f} GB/s", flush=True)
for mode in range(4):
    for per_cu in (2, 4, 8, 16, 32, 64):
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        r = timed(lambda: lib.salve_debug_copy16(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), n, per_cu * cus, mode, st))
        print(f"mode {mode} ({'nt' if mode & 1 else 'plain'}, {8 if mode & 2 else 4} in flight), {per_cu:2d} workgroups per CU: {r:.0f} GB/s", flush=True)
assert torch.equal(dst[:4096], src[:4096])

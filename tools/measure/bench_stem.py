import os
"""Stem (7x7/2 convolution + max-pool) alone at batch B (GPU box).  SALVE_RESNET_FLAGS=8: the un-fused stem; SALVE_HIP_LIB for ablation builds."""
import ctypes, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from salve_amd import _lib
from salve_amd.models import hip_resnet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
DEV = "cuda:0"
lib = _lib.load()
g = torch.Generator().manual_seed(0)
bld = hip_resnet._Builder()
w = torch.randn(64, 6, 7, 7, generator=g) * 0.05
b = torch.randn(64, generator=g) * 0.1
H, W = bld.conv(w, b, hip_resnet.NET_INPUT, 0, hip_resnet.NO_BUF, 224, 224, 2, 3, True, kw_pad=8)
bld.maxpool(0, 1, H, W, 64)
ops = np.array(bld.ops, dtype=hip_resnet.OP_DTYPE)
wb, pr, kt = np.concatenate(bld.weights).astype(np.int16), np.concatenate(bld.params).astype(np.float32), np.concatenate(bld.ktab).astype(np.int32)
h = ctypes.c_void_p(lib.salve_resnet_create(0, 8, ops.ctypes.data_as(ctypes.c_void_p), 2, wb.ctypes.data_as(ctypes.c_void_p), wb.nbytes,
                                             pr.ctypes.data_as(ctypes.c_void_p), pr.nbytes, kt.ctypes.data_as(ctypes.c_void_p), kt.size, int(os.environ.get("SALVE_RESNET_FLAGS", "0"))))
need = lib.salve_resnet_workspace_bytes(h, B)
ws = torch.zeros(need, dtype=torch.uint8, device=DEV)
x = (torch.randn(B, 224, 224, 8, generator=g) * 0.5).to(torch.float16).to(DEV)
logits = torch.zeros((B, 2), dtype=torch.float32, device=DEV)
run = lambda: lib.salve_resnet_forward(h, ctypes.c_void_p(x.data_ptr()), B, ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(), None, None)
for _ in range(3):
    assert run() == 0, lib.salve_last_error()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
per_buf = (need - 256) // 2 // 2
view = ws[(-ws.data_ptr()) % 256:].view(torch.float16)
out = view[per_buf: per_buf + B * 56 * 56 * 64].float()
print(f"stem+pool B={B}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us  sum {float(out.double().sum()):.6e} (flags={os.environ.get('SALVE_RESNET_FLAGS', '0')}, lib={os.environ.get('SALVE_HIP_LIB', 'default')})")

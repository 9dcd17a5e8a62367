import os
"""Single-convolution timings of the verifier's shapes, one configuration of the convolution kernels per run (GPU box).
usage: SALVE_RESNET_FLAGS=1|2 python tools/measure/bench_conv.py [batch]   (1 = conv_igemm_kernel only, 2 = the 8-phase kernel wherever it fits;
       SALVE_CONV_WIDE=d|e|f selects a rejected kernel in an ablation build loaded with SALVE_HIP_LIB)     prints one line per ResNet-50 shape:
time, TFLOP/s, and the checksum / max-abs of the output (to compare configurations with each other)."""
import ctypes, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from salve_amd import _lib
from salve_amd.models import hip_resnet

hip_resnet.CHUNK_MAJOR_K = os.environ.get("SALVE_K_ORDER", "") == "chunk"   # K order of the 3 x 3 shapes (hip_resnet._Builder.conv)
POWER = os.environ.get("SALVE_BENCH_POWER", "") == "1"                        # board power and clock while each shape loops for 3 s
if POWER:
    import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
DEV = "cuda:0"
lib = _lib.load()
# (name, cin, cout, k, stride, pad, hw_in, residual, src2 = (cin2, stride2, hw2))
SHAPES = [
    ("l2.conv1 1x1 512>128 @28", 512, 128, 1, 1, 0, 28, False, None),
    ("l2.conv2 3x3 128>128 @28", 128, 128, 3, 1, 1, 28, False, None),
    ("l2.conv3 1x1 128>512 +res", 128, 512, 1, 1, 0, 28, True, None),
    ("l2.0.conv2 3x3s2 128 @56", 128, 128, 3, 2, 1, 56, False, None),
    ("l2.0.conv3+sc 128|256>512", 128, 512, 1, 1, 0, 28, False, (256, 2, 56)),
    ("l3.conv1 1x1 1024>256 @14", 1024, 256, 1, 1, 0, 14, False, None),
    ("l3.conv2 3x3 256>256 @14", 256, 256, 3, 1, 1, 14, False, None),
    ("l3.conv3 1x1 256>1024 +res", 256, 1024, 1, 1, 0, 14, True, None),
    ("l3.0.conv3+sc 256|512>1024", 256, 1024, 1, 1, 0, 14, False, (512, 2, 28)),
    ("l4.conv1 1x1 2048>512 @7", 2048, 512, 1, 1, 0, 7, False, None),
    ("l4.conv2 3x3 512>512 @7", 512, 512, 3, 1, 1, 7, False, None),
    ("l4.conv3 1x1 512>2048 +res", 512, 2048, 1, 1, 0, 7, True, None),
    ("l1.conv3 1x1 64>256 @56 +res", 64, 256, 1, 1, 0, 56, True, None),
    ("l1.conv1 1x1 256>64 @56", 256, 64, 1, 1, 0, 56, False, None),
]
only = os.environ.get('SALVE_BENCH_ONLY')
if only:
    SHAPES = [sh for sh in SHAPES if any(o in sh[0] for o in only.split(','))]
reps_env = int(os.environ.get('SALVE_BENCH_REPS', '20'))
g = torch.Generator().manual_seed(0)
tot = 0.0
for name, cin, cout, k, stride, pad, hw, res, src2 in SHAPES:
    bld = hip_resnet._Builder()
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ho = (hw + 2 * pad - k) // stride + 1
    if src2 is None:
        bld.conv(w, b, hip_resnet.NET_INPUT, 0, 1 if res else hip_resnet.NO_BUF, hw, hw, stride, pad, True)
        macs = cout * cin * k * k
    else:
        cin2, s2, hw2 = src2
        w2 = torch.randn(cout, cin2, 1, 1, generator=g) * (1.0 / cin2) ** 0.5
        bld.conv1x1_with_shortcut(w, b, hip_resnet.NET_INPUT, 0, hw, hw, w2, b, 1, hw2, hw2, s2)
        macs = cout * (cin + cin2)
    ops = np.array(bld.ops, dtype=hip_resnet.OP_DTYPE)
    wb, pr, kt = np.concatenate(bld.weights).astype(np.int16), np.concatenate(bld.params).astype(np.float32), np.concatenate(bld.ktab).astype(np.int32)
    mk = lambda o, n: ctypes.c_void_p(lib.salve_resnet_create(0, cin, o.ctypes.data_as(ctypes.c_void_p), n, wb.ctypes.data_as(ctypes.c_void_p), wb.nbytes,
                                                              pr.ctypes.data_as(ctypes.c_void_p), pr.nbytes, kt.ctypes.data_as(ctypes.c_void_p), kt.size, int(os.environ.get("SALVE_RESNET_FLAGS", "0"))))
    # the handle sizes ONE activation buffer (the output); buffer 1 (residual / second source) lies behind it in the workspace
    out_elems = ho * ho * cout
    other = out_elems if src2 is None else max(out_elems, src2[2] * src2[2] * src2[0])
    need = 2 * B * (out_elems + other) + 512
    h = mk(ops, 1)
    ws = torch.zeros(need, dtype=torch.uint8, device=DEV)
    view = ws[(-ws.data_ptr()) % 256:].view(torch.float16)
    x = (torch.randn(B, hw, hw, cin, generator=g) * 0.5).to(torch.float16).to(DEV)
    gd = torch.Generator(device=DEV).manual_seed(1)
    view[: (need - 256) // 2] = (torch.randn((need - 256) // 2, generator=gd, device=DEV) * 0.5).to(torch.float16)  # residual / second source
    logits = torch.zeros((B, 2), dtype=torch.float32, device=DEV)
    run = lambda: lib.salve_resnet_forward(h, ctypes.c_void_p(x.data_ptr()), B, ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(), None, None)
    for _ in range(3):
        st = run()
    torch.cuda.synchronize()
    assert st == 0, lib.salve_last_error()
    reps = reps_env
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    out = view[: B * ho * ho * cout].float()
    flop = 2.0 * B * ho * ho * macs
    tot += us
    pw = bench.power_probe(run, 3.0) if POWER else None
    print(f"{name:30s} {us:8.1f} us  {flop / us / 1e6:7.0f} TFLOP/s   sum {float(out.double().sum()):.6e} absmax {float(out.abs().max()):.4f}"
          + (f"   {pw['power_w']:.0f} W {pw['sclk_mhz']:.0f} MHz" if pw else ""), flush=True)
    lib.salve_resnet_destroy(h)
print(f"total {tot:.1f} us  (SALVE_RESNET_FLAGS={os.environ.get('SALVE_RESNET_FLAGS', '0')})")

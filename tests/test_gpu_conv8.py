"""The 8-phase convolution kernel (salve_amd/csrc/conv8.h) against conv_igemm_kernel on single convolutions: same k order and
fp32 accumulation, so every output must agree bit for bit -- for tiles cut by the end of M, 3x3 borders, residuals and the
K-concatenated projection shortcut, and repeatedly with fresh data: its half-tile ring is ordered by counted waits and
barriers alone, a misplaced one shows up as a rare wrong tile."""
import ctypes

import numpy as np
import pytest
import torch

from salve_amd import _lib
from salve_amd.models import hip_resnet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (name, cin, cout, k, stride, pad, hw_in, residual, src2 = (cin2, stride2, hw2))
SHAPES = [
    ("1x1 1024>256 @14", 1024, 256, 1, 1, 0, 14, False, None),
    ("3x3 256>256 @14", 256, 256, 3, 1, 1, 14, False, None),
    ("3x3 512>512 @7", 512, 512, 3, 1, 1, 7, False, None),
    ("1x1 512>2048 @7 + residual", 512, 2048, 1, 1, 0, 7, True, None),
    ("1x1 256|512>1024 @14 projection", 256, 1024, 1, 1, 0, 14, False, (512, 2, 28)),
    ("3x3 / 2 256>256 @28", 256, 256, 3, 2, 1, 28, False, None),
]


def _handles(lib, monkeypatch, shape, seed, mode8):
    name, cin, cout, k, stride, pad, hw, res, src2 = shape
    g = torch.Generator().manual_seed(seed)
    bld = hip_resnet._Builder()
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    if src2 is None:
        bld.conv(w, b, hip_resnet.NET_INPUT, 0, 1 if res else hip_resnet.NO_BUF, hw, hw, stride, pad, True)
    else:
        cin2, s2, hw2 = src2
        w2 = torch.randn(cout, cin2, 1, 1, generator=g) * (1.0 / cin2) ** 0.5
        bld.conv1x1_with_shortcut(w, b, hip_resnet.NET_INPUT, 0, hw, hw, w2, b, 1, hw2, hw2, s2)
    ops = np.array(bld.ops, dtype=hip_resnet.OP_DTYPE)
    wb = np.concatenate(bld.weights).astype(np.int16)
    pr = np.concatenate(bld.params).astype(np.float32)
    kt = np.concatenate(bld.ktab).astype(np.int32)
    out = []
    for mode in ("0", mode8):
        flags = _lib.RESNET_CONV8_WHEREVER if mode == "8" else _lib.RESNET_CONV_IGEMM_ONLY
        h = lib.salve_resnet_create(0, cin, ops.ctypes.data_as(ctypes.c_void_p), 1, wb.ctypes.data_as(ctypes.c_void_p), wb.nbytes,
                                    pr.ctypes.data_as(ctypes.c_void_p), pr.nbytes, kt.ctypes.data_as(ctypes.c_void_p), kt.size, flags)
        assert h, lib.salve_last_error()
        out.append(ctypes.c_void_p(h))
    return out


@pytest.mark.parametrize("mode8", ["8"], ids=["conv8"])
@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_eight_phase_kernel_is_bit_identical_and_stays_so(monkeypatch, shape, mode8):
    lib = _lib.load()
    name, cin, cout, k, stride, pad, hw, res, src2 = shape
    ho = (hw + 2 * pad - k) // stride + 1
    h_ref, h_c8 = _handles(lib, monkeypatch, shape, seed=3, mode8=mode8)
    try:
        for B in (3, 41, 150):    # M = B ho^2: tiles cut by the end of M, one workgroup, several rounds
            out_elems = ho * ho * cout
            other = out_elems if src2 is None else max(out_elems, src2[2] * src2[2] * src2[0])
            need = 2 * B * (out_elems + other) + 512
            ws = torch.zeros(need, dtype=torch.uint8, device=DEV)
            view = ws[(-ws.data_ptr()) % 256:].view(torch.float16)
            logits = torch.zeros((B, 2), dtype=torch.float32, device=DEV)
            for rep in range(6):
                gd = torch.Generator(device=DEV).manual_seed(100 * B + rep)
                x = (torch.randn(B, hw, hw, cin, generator=gd, device=DEV) * 0.5).to(torch.float16)
                side = (torch.randn((need - 256) // 2, generator=gd, device=DEV) * 0.5).to(torch.float16)   # residual / second source
                outs = []
                for h in (h_ref, h_c8):
                    view[: (need - 256) // 2] = side
                    st = lib.salve_resnet_forward(h, ctypes.c_void_p(x.data_ptr()), B, ctypes.c_void_p(logits.data_ptr()),
                                                  ctypes.c_void_p(ws.data_ptr()), ws.numel(), None, None)
                    assert st == 0, lib.salve_last_error()
                    torch.cuda.synchronize()
                    outs.append(view[: B * out_elems].clone())
                assert torch.isfinite(outs[0].float()).all()
                assert torch.equal(outs[0], outs[1]), f"{name}: batch {B}, repetition {rep}: {int((outs[0] != outs[1]).sum())} elements differ"
    finally:
        lib.salve_resnet_destroy(h_ref)
        lib.salve_resnet_destroy(h_c8)

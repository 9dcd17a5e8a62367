"""Development: how do the rasteriser and the verifier scale with the CUs they are given?  HIP streams created with a CU mask
(hipExtStreamCreateWithCUMask) -- the question behind partitioning the chip between the two instead of time-slicing it.

    python tools/probe/cu_mask_probe.py            # per mask: verifier forward (batch 2048), scatter + densify + tiles (2048 renders)

Masks: `first K` = bits 0 .. K-1 set, `stride` = every second / fourth bit set -- the two patterns tell how the bit order maps to XCDs.
"""
import ctypes
import sys
import time
from pathlib import Path
from types import SimpleNamespace

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch

from salve_amd import synthetic
from salve_amd.models.early_fusion import EarlyFusionCEResnet
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses

hip = ctypes.CDLL("libamdhip64.so.7")   # the runtime torch has already loaded
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(dev, bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[32 * w + b]) for w in range(8)])
    s = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n = 2048
    model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    synthetic.trained_looking_batchnorm(model)
    eng = model.compiled(dev)
    x = torch.randn(n, 224, 224, eng.in_channels, device=dev).to(torch.float16)
    ras = BevRasteriser(dev)
    P = 32
    panos = [synthetic.make_pano(i) for i in range(P)]
    d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
    hyp = synthetic.make_hypotheses(n, P, seed=0)
    order = np.argsort(hyp.i1, kind="stable")
    hd = ras.upload_hypotheses(pack_hypotheses(hyp.i1[order], np.zeros(n), hyp.R[order], hyp.t[order], np.ones(n)))
    bev = torch.empty((n,) + ras.bev_hw, dtype=torch.int32, device=dev)
    ras.scatter(d_rgb, d_depth, hd, n, bev); ras.densify(n, bev); eng.forward_nhwc(x)   # warm-up on the default stream (index build, workspaces)
    torch.cuda.synchronize()

    def timed(stream, fn, reps=3):
        with torch.cuda.stream(stream):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                fn()
            e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    masks = {"all 256": [1] * 256}
    for k in (192, 128, 96, 64, 32):
        masks[f"first {k}"] = [1 if b < k else 0 for b in range(256)]
    masks["every 2nd (128)"] = [1 if b % 2 == 0 else 0 for b in range(256)]
    masks["every 4th (64)"] = [1 if b % 4 == 0 else 0 for b in range(256)]
    masks["last 64"] = [1 if b >= 192 else 0 for b in range(256)]
    print(f"{'mask':18s} {'verifier ms':>12s} {'scatter ms':>11s} {'densify ms':>11s}   (batch / renders {n})", flush=True)
    for name, bits in masks.items():
        s = masked_stream(dev, bits)
        tv = timed(s, lambda: eng.forward_nhwc(x))
        ts = timed(s, lambda: ras.scatter(d_rgb, d_depth, hd, n, bev))
        td = timed(s, lambda: ras.densify(n, bev))
        print(f"{name:18s} {tv:12.2f} {ts:11.2f} {td:11.2f}", flush=True)
    # concurrency: verifier on the first 176 CUs, rasteriser on the last 80, at the same time
    for kv in (192, 176, 160):
        sv = masked_stream(dev, [1 if b < kv else 0 for b in range(256)])
        sr = masked_stream(dev, [1 if b >= kv else 0 for b in range(256)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            with torch.cuda.stream(sv):
                eng.forward_nhwc(x)
            with torch.cuda.stream(sr):
                ras.scatter(d_rgb, d_depth, hd, n, bev); ras.densify(n, bev)
        torch.cuda.synchronize()
        print(f"concurrent: verifier on first {kv}, rasteriser on last {256 - kv}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per (forward + scatter + densify) of {n}", flush=True)
    from salve_amd import status
    status.check(dev, "cu_mask_probe")


if __name__ == "__main__":
    main()

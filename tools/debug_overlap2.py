import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from _helpers import load_testhelp
helper = load_testhelp()   # the synthetic load kernels live in the test helper library, not in the product
dev = torch.device("cuda:0")
ras = BevRasteriser(dev)
panos = [synthetic.make_pano(i) for i in range(4)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
n = 64
hyp = synthetic.make_hypotheses(n, 4, seed=1)
hd = ras.upload_hypotheses(pack_hypotheses(hyp.i1, np.zeros(n), hyp.R, hyp.t, np.ones(n)))
bev0, _ = ras.render(d_rgb, d_depth, hd, n); torch.cuda.synchronize(); ref = bev0.clone()
A = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "mm"
for rep in range(6):
    with torch.cuda.stream(side):
        for _ in range(3):
            if mode == "mm":
                B = A @ A
            elif mode == "copy":
                B = A.clone()
            elif mode == "small":
                for _ in range(300):
                    A[:4096].add_(1)
    out, _ = ras.render(d_rgb, d_depth, hd, n)
    torch.cuda.synchronize()
    d = (out != ref).reshape(n, -1).any(1).nonzero().flatten().tolist()
    print(mode, "rep", rep, "renders differing:", d)
if mode.startswith("resnet"):
    pass
if mode in ("resnet", "resnet_stray", "resnet_bisect"):
    from types import SimpleNamespace
    from salve_amd.models.early_fusion import EarlyFusionCEResnet
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"])).eval()
    eng = model.compiled(dev)
    x = torch.randn(64, 224, 224, 8, device=dev).to(torch.float16)
    y0 = eng.forward_nhwc(x).clone(); torch.cuda.synchronize()
    for rep in range(6 if mode == "resnet" else 0):
        with torch.cuda.stream(side):
            for _ in range(2):
                y = eng.forward_nhwc(x)
        out, _ = ras.render(d_rgb, d_depth, hd, n)
        torch.cuda.synchronize()
        d = (out != ref).reshape(n, -1).any(1).nonzero().flatten().tolist()
        print(mode, "rep", rep, "renders differing:", d, "logits equal", bool(torch.equal(y, y0)))
if mode == "resnet_stray":
    ras.scatter(d_rgb, d_depth, hd, n); torch.cuda.synchronize()
    snap_ws = ras._ws.clone(); snap_bev = ref.clone()
    others = {"d_rgb": d_rgb.clone(), "hd": hd.clone(), "sphere": ras.sphere.clone()}
    for _ in range(5):
        y = eng.forward_nhwc(x)
    torch.cuda.synchronize()
    diff = (ras._ws != snap_ws).nonzero().flatten()
    print("workspace bytes changed by the ResNet forward:", diff.numel(), diff[:10].tolist())
    print("ref bev changed:", int((ref != snap_bev).sum()), "rgb changed", int((d_rgb != others["d_rgb"]).sum()))
    print("ws ptr", hex(ras._ws.data_ptr()), "size", ras._ws.numel(), "resnet ws ptr", hex(eng._ws.data_ptr()), "size", eng._ws.numel(), "x ptr", hex(x.data_ptr()))
if mode == "resnet_bisect":
    out0, dbg0 = ras.render(d_rgb, d_depth, hd, n, debug=True); torch.cuda.synchronize()
    k0, s0 = dbg0.keys.clone(), dbg0.stats.clone()
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(2):
                y = eng.forward_nhwc(x)
        out, dbg = ras.render(d_rgb, d_depth, hd, n, debug=True)
        torch.cuda.synchronize()
        dk = (dbg.keys != k0).reshape(n, -1).any(1).nonzero().flatten().tolist()
        db = (out != ref).reshape(n, -1).any(1).nonzero().flatten().tolist()
        ds = (dbg.stats != s0).any(1).nonzero().flatten().tolist()
        print("rep", rep, "keys differ:", dk, "bev differ:", db, "stats differ:", ds, [dbg.stats[i].tolist() for i in ds[:2]], [s0[i].tolist() for i in ds[:2]])
if mode.startswith("op_"):
    # run a single-op "network" concurrently: op_conv1x1, op_conv3x3, op_maxpool
    import ctypes
    from salve_amd import _lib
    from salve_amd.models import hip_resnet as hr
    lib = _lib.load()
    bld = hr._Builder()
    g = torch.Generator().manual_seed(0)
    B = 64
    if mode == "op_maxpool":
        bld.maxpool(hr.NET_INPUT, 0, 112, 112, 64); cin = 64; hw = 112
        bld.weights.append(np.zeros(8, np.int16)); bld.params.append(np.zeros(8, np.float32)); bld.ktab.append(np.zeros(8, np.int32))
    else:
        k = 1 if mode == "op_conv1x1" else 3
        cin, hw = 256, 56
        bld.conv(torch.randn(256, cin, k, k, generator=g) * 0.05, torch.zeros(256), hr.NET_INPUT, 0, hr.NO_BUF, hw, hw, 1, k // 2, True)
    ops = np.array(bld.ops, dtype=hr.OP_DTYPE)
    wb, pr, kt = np.concatenate(bld.weights).astype(np.int16), np.concatenate(bld.params).astype(np.float32), np.concatenate(bld.ktab).astype(np.int32)
    h = ctypes.c_void_p(lib.salve_resnet_create(0, cin, ops.ctypes.data_as(ctypes.c_void_p), len(ops), wb.ctypes.data_as(ctypes.c_void_p), wb.nbytes,
                                                pr.ctypes.data_as(ctypes.c_void_p), pr.nbytes, kt.ctypes.data_as(ctypes.c_void_p), kt.size, 0))
    ws = torch.zeros(lib.salve_resnet_workspace_bytes(h, B), dtype=torch.uint8, device=dev)
    xin = torch.randn(B, hw, hw, cin, device=dev).to(torch.float16)
    logits = torch.zeros(B, 2, device=dev)
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(40):
                lib.salve_resnet_forward(h, ctypes.c_void_p(xin.data_ptr()), B, ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                         None, ctypes.c_void_p(side.cuda_stream))
        out, _ = ras.render(d_rgb, d_depth, hd, n)
        torch.cuda.synchronize()
        d = (out != ref).reshape(n, -1).any(1).nonzero().flatten().tolist()
        print(mode, "rep", rep, "renders differing:", d)

if mode.startswith("burn") and mode[4:].isdigit():
    import ctypes
    from salve_amd import _lib
    lib = _lib.load()
    bm = int(mode[4:])
    sink = torch.zeros(4, device=dev)
    for rep in range(6):
        for _ in range(6):
            helper.salve_debug_burn(1024, 4000, bm, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream))
        out, _ = ras.render(d_rgb, d_depth, hd, n)
        torch.cuda.synchronize()
        d = (out != ref).reshape(n, -1).any(1).nonzero().flatten().tolist()
        print(mode, "rep", rep, "renders differing:", d)
if mode == "burnstats":
    import ctypes
    from salve_amd import _lib
    lib = _lib.load()
    sink = torch.zeros(4, device=dev)
    out0, dbg0 = ras.render(d_rgb, d_depth, hd, n, debug=True); torch.cuda.synchronize()
    for rep in range(3):
        for _ in range(6):
            helper.salve_debug_burn(1024, 4000, 0, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream))
        out, dbg = ras.render(d_rgb, d_depth, hd, n, debug=True)
        torch.cuda.synchronize()
        print("rep", rep, "keys differ", int((dbg.keys != dbg0.keys).sum()), "mask differ", int((dbg.mask != dbg0.mask).sum()),
              "img_xy differ", int((dbg.img_xy != dbg0.img_xy).sum()), "bev px differ", int((out != out0).sum()))
        ds = (dbg.stats != dbg0.stats).any(1).nonzero().flatten().tolist()
        print("   stats differ in", len(ds), "renders; e.g.", dbg.stats[ds[0]].tolist() if ds else None, dbg0.stats[ds[0]].tolist() if ds else None)

"""GPU parity of the HIP ResNet verifier (fp16 MFMA) against the fp32 CPU oracle, through the C ABI."""

import ctypes
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import resnet_oracle as ro  # noqa: E402
from salve_amd import _lib  # noqa: E402
from salve_amd.models import hip_resnet  # noqa: E402
from salve_amd.models.early_fusion import EarlyFusionCEResnet  # noqa: E402

DEV = "cuda:0"


def randomise_bn(model, seed=0):
    """Trained-looking BatchNorm statistics so that activations stay O(1) through the trunk."""
    g = torch.Generator().manual_seed(seed)
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            last = name.endswith("bn3") or (name.endswith("bn2") and model.resnet.block_kind == "basic")
            m.weight.data = (0.25 if last else 1.0) * (0.6 + 0.4 * torch.rand(m.num_features, generator=g))
            m.bias.data = 0.1 * torch.randn(m.num_features, generator=g)
            m.running_mean.data = 0.1 * torch.randn(m.num_features, generator=g)
            m.running_var.data = 0.6 + 0.8 * torch.rand(m.num_features, generator=g)


def run_single_conv(w, b, x_nhwc, stride, pad, relu, res=None, kw_pad=0):
    """One CONV op through salve_resnet_create / salve_resnet_forward; returns the fp16 NHWC output as fp32."""
    lib = _lib.load()
    bld = hip_resnet._Builder()
    B, Hi, Wi, Cp = x_nhwc.shape
    Ho, Wo = bld.conv(w, b, hip_resnet.NET_INPUT, 0, hip_resnet.NO_BUF if res is None else 1, Hi, Wi, stride, pad, relu, kw_pad)
    ops = np.array(bld.ops, dtype=hip_resnet.OP_DTYPE)
    wb, pr, kt = np.concatenate(bld.weights).astype(np.int16), np.concatenate(bld.params).astype(np.float32), np.concatenate(bld.ktab).astype(np.int32)
    if res is not None:  # make the handle size two buffers
        ops = np.concatenate([ops, ops])
        ops[1]["out_buf"] = 1
        ops = ops[:1] if False else ops
    h = lib.salve_resnet_create(0, Cp, ops.ctypes.data_as(ctypes.c_void_p), len(ops), wb.ctypes.data_as(ctypes.c_void_p), wb.nbytes,
                                pr.ctypes.data_as(ctypes.c_void_p), pr.nbytes, kt.ctypes.data_as(ctypes.c_void_p), kt.size)
    assert h
    h = ctypes.c_void_p(h)
    need = lib.salve_resnet_workspace_bytes(h, B)
    ws = torch.zeros(need, dtype=torch.uint8, device=DEV)
    Cout = w.shape[0]
    per_buf = (need - 256) // (2 if res is not None else 1) // 2
    base_off = (-ws.data_ptr()) % 256
    view = ws[base_off:].view(torch.float16)
    if res is not None:
        view[per_buf:per_buf + res.numel()] = res.reshape(-1).to(DEV)
        ops1 = ops[:1]
        lib.salve_resnet_destroy(h)
        h = ctypes.c_void_p(lib.salve_resnet_create(0, Cp, ops.ctypes.data_as(ctypes.c_void_p), 1, wb.ctypes.data_as(ctypes.c_void_p), wb.nbytes,
                                                     pr.ctypes.data_as(ctypes.c_void_p), pr.nbytes, kt.ctypes.data_as(ctypes.c_void_p), kt.size))
        # one op, but the workspace keeps room for two buffers (the residual lives in buffer 1)
    logits = torch.zeros((B, 2), dtype=torch.float32, device=DEV)
    xd = x_nhwc.to(DEV).contiguous()
    st = lib.salve_resnet_forward(h, ctypes.c_void_p(xd.data_ptr()), B, ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                  ws.numel(), None)
    torch.cuda.synchronize()
    assert st == 0, lib.salve_last_error()
    out = view[: B * Ho * Wo * Cout].float().cpu().reshape(B, Ho, Wo, Cout)
    lib.salve_resnet_destroy(h)
    return out


@pytest.mark.parametrize("case", [
    dict(cin=64, cout=64, k=1, s=1, p=0, hw=56, relu=True),
    dict(cin=64, cout=256, k=1, s=1, p=0, hw=56, relu=False),
    dict(cin=128, cout=128, k=3, s=2, p=1, hw=28, relu=True),
    dict(cin=256, cout=512, k=1, s=2, p=0, hw=28, relu=False),
    dict(cin=512, cout=512, k=3, s=1, p=1, hw=7, relu=True),
    dict(cin=6, cout=64, k=7, s=2, p=3, hw=64, relu=True, kw_pad=8),
    dict(cin=12, cout=64, k=7, s=2, p=3, hw=32, relu=True, kw_pad=8),
])
def test_conv_matches_torch(case):
    g = torch.Generator().manual_seed(1)
    B = 3
    w = torch.randn(case["cout"], case["cin"], case["k"], case["k"], generator=g) * (2.0 / (case["cin"] * case["k"] ** 2)) ** 0.5
    b = torch.randn(case["cout"], generator=g) * 0.1
    x = torch.randn(B, case["cin"], case["hw"], case["hw"], generator=g)
    cp = hip_resnet.pad_channels(case["cin"])
    x_nhwc = torch.zeros(B, case["hw"], case["hw"], cp, dtype=torch.float16)
    x_nhwc[..., : case["cin"]] = x.permute(0, 2, 3, 1).to(torch.float16)
    got = run_single_conv(w, b, x_nhwc, case["s"], case["p"], case["relu"], kw_pad=case.get("kw_pad", 0))
    xr = x_nhwc[..., : case["cin"]].float().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xr, w.to(torch.float16).float(), b, case["s"], case["p"])
    if case["relu"]:
        ref = ref.relu()
    ref = ref.permute(0, 2, 3, 1)
    err = (got - ref).abs()
    tol = 3e-3 * ref.abs().clamp(min=0.5)  # three fp16 ulps of the output (fp32 accumulation, one rounding at the store)
    assert (err <= tol).all(), f"max err {err.max()} at |ref| {ref.abs().max()}"


@pytest.mark.parametrize("num_layers,modalities,batch", [
    (50, ["floor_rgb_texture"], 5),
    (152, ["ceiling_rgb_texture", "floor_rgb_texture"], 2),
    (18, ["layout"], 3),
])
def test_logits_match_oracle(num_layers, modalities, batch):
    """north_star asks 1e-3 on the logits.  Activations and weights are fp16 (11 significand bits), accumulation fp32:
    the bound here is 1e-3 of the logit scale (>= 1), and the measured error is printed."""
    torch.manual_seed(0)
    model = EarlyFusionCEResnet(num_layers, False, 2, SimpleNamespace(modalities=modalities))
    randomise_bn(model)
    model.eval()
    n = len(modalities) * 2
    xs = [torch.randn(batch, 3, 224, 224) for _ in range(n)]
    xs_b = [x.to(torch.float16).float() for x in xs]  # the network input is fp16 on the GPU side
    with torch.no_grad():
        ref = ro.forward(model.state_dict(), num_layers, xs_b)
        pad = xs + [None] * (6 - n)
        got = model.to(DEV)(*[None if x is None else x.to(DEV) for x in pad]).cpu()
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    print(f"resnet{num_layers}: logits scale {scale:.3f}, max abs err {err:.4f}")
    assert err <= 1e-3 * scale
    assert (got.argmax(1) == ref.argmax(1)).all() or err < 1e-3


@pytest.mark.parametrize("stride,hw,cx,mid,cout", [(1, 56, 64, 64, 256), (2, 28, 256, 128, 512)])
def test_conv_with_projection_shortcut_as_second_source(stride, hw, cx, mid, cout):
    """`in2_buf` of salve_resnet_op_t: relu(W . t2 + b + W2 . x[::s, ::s] + b2) as one GEMM over concatenated channels,
    against torch (the down-sampling bottleneck's last convolution + projection shortcut)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(7)
    B = 2
    wb3 = torch.randn(mid, cx, 3, 3, generator=g) * (2.0 / (cx * 9)) ** 0.5
    bb3 = torch.randn(mid, generator=g) * 0.1
    w = torch.randn(cout, mid, 1, 1, generator=g) * (1.0 / mid) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    w2 = torch.randn(cout, cx, 1, 1, generator=g) * (1.0 / cx) ** 0.5
    b2 = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, cx, hw, hw, generator=g)
    x_nhwc = x.permute(0, 2, 3, 1).to(torch.float16).contiguous()
    bld = hip_resnet._Builder()
    Ho, Wo = bld.conv(wb3, bb3, hip_resnet.NET_INPUT, 0, hip_resnet.NO_BUF, hw, hw, stride, 1, True)
    bld.conv1x1_with_shortcut(w, b, 0, 1, Ho, Wo, w2, b2, hip_resnet.NET_INPUT, hw, hw, stride)
    ops = np.array(bld.ops, dtype=hip_resnet.OP_DTYPE)
    wts, pr, kt = np.concatenate(bld.weights).astype(np.int16), np.concatenate(bld.params).astype(np.float32), np.concatenate(bld.ktab).astype(np.int32)
    h = ctypes.c_void_p(lib.salve_resnet_create(0, cx, ops.ctypes.data_as(ctypes.c_void_p), len(ops), wts.ctypes.data_as(ctypes.c_void_p), wts.nbytes,
                                                 pr.ctypes.data_as(ctypes.c_void_p), pr.nbytes, kt.ctypes.data_as(ctypes.c_void_p), kt.size))
    assert h
    need = lib.salve_resnet_workspace_bytes(h, B)
    ws = torch.zeros(need, dtype=torch.uint8, device=DEV)
    xd = x_nhwc.to(DEV)
    logits = torch.zeros((B, 2), dtype=torch.float32, device=DEV)
    st = lib.salve_resnet_forward(h, ctypes.c_void_p(xd.data_ptr()), B, ctypes.c_void_p(logits.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws.numel(), None)
    torch.cuda.synchronize()
    assert st == 0, lib.salve_last_error()
    per_buf = (need - 256) // 2 // 2                    # two buffers of 16-bit elements
    view = ws[(-ws.data_ptr()) % 256:].view(torch.float16)
    got = view[per_buf: per_buf + B * Ho * Wo * cout].float().cpu().reshape(B, Ho, Wo, cout)
    lib.salve_resnet_destroy(h)
    xr = x_nhwc.float().permute(0, 3, 1, 2)
    t2 = torch.nn.functional.conv2d(xr, wb3.to(torch.float16).float(), bb3, stride, 1).relu().to(torch.float16).float()
    ref = (torch.nn.functional.conv2d(t2, w.to(torch.float16).float(), b) +
           torch.nn.functional.conv2d(xr, w2.to(torch.float16).float(), b2, stride)).relu().permute(0, 2, 3, 1)
    err = (got - ref).abs()
    assert (err <= 3e-3 * ref.abs().clamp(min=0.5)).all(), f"max err {err.max()}"


def test_fused_bottleneck_is_bit_identical_to_three_kernels(monkeypatch):
    """The fused bottleneck kernel (resnet.hip: bottleneck_kernel) keeps t1 / t2 in LDS but rounds them to fp16 and
    accumulates in the same k order as the three separate convolutions: the logits must agree bit for bit."""
    torch.manual_seed(5)
    model = EarlyFusionCEResnet(50, False, 2, SimpleNamespace(modalities=["floor_rgb_texture"]))
    randomise_bn(model, seed=5)
    model.eval()
    x = torch.randn(3, 224, 224, 8).to(torch.float16).to(DEV)
    x[..., 6:] = 0
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("SALVE_RESNET_FUSE", fuse)   # read when the handle is created
        eng = hip_resnet.HipResNet(model.state_dict(), 50, torch.device(DEV))
        outs.append(eng.forward_nhwc(x).clone())
        torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])


def test_forward_refuses_cpu_and_bad_modalities():
    model = EarlyFusionCEResnet(18, False, 2, SimpleNamespace(modalities=["layout"])).eval()
    x = torch.zeros(1, 3, 224, 224)
    with pytest.raises(RuntimeError):
        model(x, x, None, None, None, None)
    with pytest.raises(RuntimeError):
        EarlyFusionCEResnet(18, False, 2, SimpleNamespace(modalities=["bogus"]))

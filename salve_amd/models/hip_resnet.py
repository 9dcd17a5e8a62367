"""Host side of the HIP verifier: fold BatchNorm, pack weights, emit the op program, run it.

No arithmetic of the forward pass happens here -- this module only prepares the immutable weight blobs
(once per checkpoint) and owns the workspace; the network runs in salve_amd/csrc/resnet.hip.
"""

from __future__ import annotations

import ctypes
from typing import Dict, List, Tuple

import numpy as np
import torch

from salve_amd import _lib, status
from salve_amd.models.resnet_factory import RESNET_SPECS

OP_CONV, OP_MAXPOOL, OP_AVGPOOL_FC = 0, 1, 2
NET_INPUT, NO_BUF = -1, -2
BN_EPS = 1e-5

OP_DTYPE = np.dtype(
    [("op", "<i4"), ("in_buf", "<i4"), ("out_buf", "<i4"), ("res_buf", "<i4"), ("Hi", "<i4"), ("Wi", "<i4"),
     ("Cin", "<i4"), ("Ho", "<i4"), ("Wo", "<i4"), ("Cout", "<i4"), ("KH", "<i4"), ("KW", "<i4"), ("stride", "<i4"),
     ("pad", "<i4"), ("relu", "<i4"), ("reserved", "<i4"), ("w_off", "<i8"), ("b_off", "<i8"), ("ktab_off", "<i8"),
     ("in2_buf", "<i4"), ("Cin2", "<i4"), ("stride2", "<i4"), ("Hi2", "<i4"), ("Wi2", "<i4"), ("reserved2", "<i4")]
)
assert OP_DTYPE.itemsize == 112


def pad_channels(c: int) -> int:
    return (c + 7) // 8 * 8


def fold_bn(w: torch.Tensor, bn: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """conv (no bias) followed by eval-mode BatchNorm == conv with scaled weights and a bias (fp32)."""
    scale = bn["weight"].float() / torch.sqrt(bn["running_var"].float() + BN_EPS)
    return w.float() * scale[:, None, None, None], bn["bias"].float() - bn["running_mean"].float() * scale


class _Builder:
    def __init__(self) -> None:
        self.ops: List[tuple] = []
        self.weights: List[np.ndarray] = []
        self.params: List[np.ndarray] = []
        self.ktab: List[np.ndarray] = []
        self.w_elems = 0
        self.p_elems = 0
        self.k_elems = 0

    def conv(self, w: torch.Tensor, b: torch.Tensor, in_buf: int, out_buf: int, res_buf: int, Hi: int, Wi: int,
             stride: int, pad: int, relu: bool, kw_pad: int = 0) -> Tuple[int, int]:
        Cout, Cin, KH, KW = w.shape
        Cinp = pad_channels(Cin)
        KWp = kw_pad or KW
        wp = torch.zeros((Cout, KH, KWp, Cinp), dtype=torch.float32)
        wp[:, :, :KW, :Cin] = w.permute(0, 2, 3, 1)
        K = KH * KWp * Cinp
        assert K % 64 == 0 and Cout % 64 == 0, (K, Cout)
        q = np.arange(K // 8)
        if KH == 7 and KWp == 8 and Cinp in (16, 24):
            # the 16- / 24-channel stem (two surfaces: 12 channels; + layout: 18): K ordered (group of 8 channels, kh, kw, channel in group), so that a k-tile of 64 is one
            # kernel row of ONE channel group -- the order in which the fused stem kernel (stem_pool.h, G = 2 / 3) walks its channel groups; the implicit-GEMM
            # kernel follows the k table, so the two-kernel path multiplies in the same order (bit-identical pooled tensors)
            wp = wp.reshape(Cout, KH, KWp, Cinp // 8, 8).permute(0, 3, 1, 2, 4).contiguous()
            c8 = q // (KH * KWp)
            kh = (q % (KH * KWp)) // KWp
            kw = q % KWp
        else:
            c8 = q % (Cinp // 8)
            kw = (q // (Cinp // 8)) % KWp
            kh = q // ((Cinp // 8) * KWp)
        bits = wp.reshape(Cout, K).to(torch.float16).view(torch.int16).numpy().copy()
        tab = ((kh & 0xFF) | ((kw & 0xFF) << 8) | ((c8 * 8) << 16)).astype(np.int32)
        Ho = (Hi + 2 * pad - KH) // stride + 1
        Wo = (Wi + 2 * pad - KW) // stride + 1
        self.ops.append((OP_CONV, in_buf, out_buf, res_buf, Hi, Wi, Cinp, Ho, Wo, Cout, KH, KWp, stride, pad, int(relu), 0,
                         self.w_elems, self.p_elems, self.k_elems, NO_BUF, 0, 0, 0, 0, 0))
        self.weights.append(bits.reshape(-1))
        self.params.append(b.float().numpy().copy())
        self.ktab.append(tab)
        self.w_elems += bits.size
        self.p_elems += Cout
        self.k_elems += tab.size
        return Ho, Wo

    def conv1x1_with_shortcut(self, w: torch.Tensor, b: torch.Tensor, in_buf: int, out_buf: int, Hi: int, Wi: int,
                              w2: torch.Tensor, b2: torch.Tensor, in2_buf: int, Hi2: int, Wi2: int, stride2: int) -> None:
        """relu(w . in + b + w2 . in2[::stride2, ::stride2] + b2) as ONE op: the last 1x1 convolution of a down-sampling
        bottleneck block with the block's projection shortcut folded in (include/salve_hip.h, `in2_buf`).  The weight rows
        are the concatenation [w | w2]; the two biases add."""
        Cout, Cin = w.shape[:2]
        Cin2 = w2.shape[1]
        assert w.shape[2:] == (1, 1) and w2.shape[2:] == (1, 1) and Cin % 64 == 0 and Cin2 % 64 == 0 and Cout % 64 == 0
        assert (Hi2 - 1) // stride2 + 1 == Hi and (Wi2 - 1) // stride2 + 1 == Wi
        wcat = torch.cat([w.reshape(Cout, Cin), w2.reshape(Cout, Cin2)], dim=1).float()
        bits = wcat.to(torch.float16).view(torch.int16).numpy().copy()
        tab = np.zeros((Cin + Cin2) // 8, dtype=np.int32)  # not read by the point-wise kernel
        self.ops.append((OP_CONV, in_buf, out_buf, NO_BUF, Hi, Wi, Cin, Hi, Wi, Cout, 1, 1, 1, 0, 1, 0,
                         self.w_elems, self.p_elems, self.k_elems, in2_buf, Cin2, stride2, Hi2, Wi2, 0))
        self.weights.append(bits.reshape(-1))
        self.params.append((b.float() + b2.float()).numpy().copy())
        self.ktab.append(tab)
        self.w_elems += bits.size
        self.p_elems += Cout
        self.k_elems += tab.size

    def maxpool(self, in_buf: int, out_buf: int, Hi: int, Wi: int, C: int) -> Tuple[int, int]:
        Ho, Wo = (Hi + 2 - 3) // 2 + 1, (Wi + 2 - 3) // 2 + 1
        self.ops.append((OP_MAXPOOL, in_buf, out_buf, NO_BUF, Hi, Wi, C, Ho, Wo, C, 3, 3, 2, 1, 0, 0, 0, 0, 0, NO_BUF, 0, 0, 0, 0, 0))
        return Ho, Wo

    def fc(self, w: torch.Tensor, b: torch.Tensor, in_buf: int, Hi: int, Wi: int, C: int) -> None:
        ncls = w.shape[0]
        w_off = self.p_elems
        self.params.append(w.float().numpy().reshape(-1).copy())
        self.p_elems += w.numel()
        b_off = self.p_elems
        self.params.append(b.float().numpy().copy())
        self.p_elems += ncls
        self.ops.append((OP_AVGPOOL_FC, in_buf, NO_BUF, NO_BUF, Hi, Wi, C, 1, 1, ncls, 1, 1, 1, 0, 0, 0, w_off, b_off, 0, NO_BUF, 0, 0, 0, 0, 0))


def _bn(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    return {k: sd[f"{prefix}.{k}"] for k in ("weight", "bias", "running_mean", "running_var")}


def build_program(state_dict: Dict[str, torch.Tensor], num_layers: int, in_hw: Tuple[int, int] = (224, 224)):
    """state_dict with the reference's keys (conv1.weight, fc.*, resnet.*; an optional `module.` prefix from
    DataParallel is stripped) -> (ops array, fp16 weight bits, fp32 params, ktab, padded input channels)."""
    sd = {(k[len("module."):] if k.startswith("module.") else k): v.detach().cpu() for k, v in state_dict.items()}
    kind, blocks = RESNET_SPECS[num_layers]
    b = _Builder()
    H, W = in_hw
    w1, bias1 = fold_bn(sd["conv1.weight"], _bn(sd, "resnet.bn1"))
    cin_p = pad_channels(w1.shape[1])
    H, W = b.conv(w1, bias1, NET_INPUT, 0, NO_BUF, H, W, stride=2, pad=3, relu=True, kw_pad=8)
    H, W = b.maxpool(0, 1, H, W, 64)
    x = 1
    for si, n in enumerate(blocks):
        for bi in range(n):
            p = f"resnet.layer{si + 1}.{bi}"
            stride = 2 if (bi == 0 and si > 0) else 1
            free = [i for i in range(5) if i != x]
            t1, t2, dsb, outb = free[0], free[1], free[2], free[3]
            if kind == "bottleneck":
                wa, ba = fold_bn(sd[f"{p}.conv1.weight"], _bn(sd, f"{p}.bn1"))
                wb, bb = fold_bn(sd[f"{p}.conv2.weight"], _bn(sd, f"{p}.bn2"))
                wc, bc = fold_bn(sd[f"{p}.conv3.weight"], _bn(sd, f"{p}.bn3"))
                b.conv(wa, ba, x, t1, NO_BUF, H, W, 1, 0, True)
                Ho, Wo = b.conv(wb, bb, t1, t2, NO_BUF, H, W, stride, 1, True)
                if f"{p}.downsample.0.weight" in sd:
                    # the projection shortcut rides along in the last convolution (one GEMM over [t2 | x] channels)
                    wd, bd = fold_bn(sd[f"{p}.downsample.0.weight"], _bn(sd, f"{p}.downsample.1"))
                    b.conv1x1_with_shortcut(wc, bc, t2, outb, Ho, Wo, wd, bd, x, H, W, stride)
                else:
                    b.conv(wc, bc, t2, outb, x, Ho, Wo, 1, 0, True)
            else:
                wa, ba = fold_bn(sd[f"{p}.conv1.weight"], _bn(sd, f"{p}.bn1"))
                wb, bb = fold_bn(sd[f"{p}.conv2.weight"], _bn(sd, f"{p}.bn2"))
                Ho, Wo = b.conv(wa, ba, x, t1, NO_BUF, H, W, stride, 1, True)
                idn = x
                if f"{p}.downsample.0.weight" in sd:
                    wd, bd = fold_bn(sd[f"{p}.downsample.0.weight"], _bn(sd, f"{p}.downsample.1"))
                    b.conv(wd, bd, x, dsb, NO_BUF, H, W, stride, 0, False)
                    idn = dsb
                b.conv(wb, bb, t1, outb, idn, Ho, Wo, 1, 1, True)
            x, H, W = outb, Ho, Wo
    feat = sd["fc.weight"].shape[1]
    b.fc(sd["fc.weight"], sd["fc.bias"], x, H, W, feat)
    ops = np.array(b.ops, dtype=OP_DTYPE)
    # the kernels' ReLUs and range tracking swallow NaNs (resnet.hip: track4): a non-finite weight -- a diverged checkpoint --
    # would come out as finite, wrong logits; refuse it here, where torch would have propagated it to the output
    for arr, what in ((np.concatenate(b.weights).astype(np.int16).view(np.float16), "convolution weights (after folding BatchNorm)"),
                      (np.concatenate(b.params), "biases / fc parameters")):
        if not np.isfinite(arr.astype(np.float32)).all():
            raise ValueError(f"checkpoint has non-finite {what}, or values beyond the fp16 range")
    return ops, np.concatenate(b.weights).astype(np.int16), np.concatenate(b.params).astype(np.float32), \
        np.concatenate(b.ktab).astype(np.int32), cin_p


class HipResNet:
    """A compiled verifier on one GPU: immutable device weights inside a library handle + a workspace."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], num_layers: int, device: torch.device, flags: int = 0) -> None:
        """flags: _lib.RESNET_* kernel selection (0 = the product's; the bit-identity tests pass the others)."""
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SalveHipError("HipResNet needs a HIP device ('cuda:N'); there is no CPU path")
        ops, wbits, params, ktab, cin_p = build_program(state_dict, num_layers)
        self.in_channels = cin_p
        self.num_classes = int(ops[-1]["Cout"])
        self.n_ops = len(ops)
        with torch.cuda.device(self.device):
            h = self.lib.salve_resnet_create(
                num_layers, cin_p, ops.ctypes.data_as(ctypes.c_void_p), len(ops),
                wbits.ctypes.data_as(ctypes.c_void_p), wbits.nbytes, params.ctypes.data_as(ctypes.c_void_p), params.nbytes,
                ktab.ctypes.data_as(ctypes.c_void_p), ktab.size, int(flags),
            )
        if not h:
            _lib.check(-1, "salve_resnet_create")
        self.handle = ctypes.c_void_p(h)
        self._ws = None

    def __del__(self) -> None:
        try:
            if getattr(self, "handle", None):
                self.lib.salve_resnet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def forward_nhwc(self, x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """x: fp16 [B, 224, 224, in_channels] on the device -> fp32 logits [B, num_classes]."""
        assert x.dtype == torch.float16 and x.is_contiguous() and x.shape[-1] == self.in_channels, (x.dtype, x.shape)
        B = int(x.shape[0])
        need = self.lib.salve_resnet_workspace_bytes(self.handle, B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty((B, self.num_classes), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            st = self.lib.salve_resnet_forward(
                self.handle, ctypes.c_void_p(x.data_ptr()), B, ctypes.c_void_p(out.data_ptr()),
                ctypes.c_void_p(self._ws.data_ptr()), self._ws.numel(), status.ptr(self.device),
                ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream),
            )
        _lib.check(st, "salve_resnet_forward")
        return out


def nchw_to_input(xs: List[torch.Tensor], in_channels: int) -> torch.Tensor:
    """[B,3,H,W] fp32 tensors (the reference's x1..x6) -> one fp16 NHWC tensor padded to in_channels."""
    x = torch.cat(xs, dim=1)
    B, C, H, W = x.shape
    out = torch.zeros((B, H, W, in_channels), dtype=torch.float16, device=x.device)
    out[..., :C] = x.permute(0, 2, 3, 1).to(torch.float16)
    return out

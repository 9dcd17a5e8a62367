"""Device status word shared by the launches of one GPU (include/salve_hip.h: SALVE_STATUS_*).

Kernels OR bits into it (a Delaunay star walk that did not close, an activation beyond the fp16 range); the launches are
asynchronous, so the host reads the word at a point where it synchronises anyway -- `check(device, what)` -- and raises.
"""

from __future__ import annotations

import ctypes
from typing import Dict

import torch

from salve_amd import _lib

_words: Dict[str, torch.Tensor] = {}


def word(device) -> torch.Tensor:
    device = torch.device(device)
    key = str(device) if device.index is not None else f"{device.type}:{torch.cuda.current_device()}"
    t = _words.get(key)
    if t is None:
        t = _words[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(word(device).data_ptr())


def check(device, what: str) -> None:
    """Synchronising read of the status word; a non-zero word is reset and raised as SalveHipError."""
    t = word(device)
    v = int(t.item())
    if v:
        t.zero_()
        _lib.check_status_word(v, what)

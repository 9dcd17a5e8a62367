"""roctx ranges around the stages of the hot path (the reference has no tracing; the build adds ranges
so that `rocprofv3 --marker-trace` attributes host-side time per stage).  `libroctx64.so` is resolved at first use; when
it is absent the ranges are no-ops."""

from __future__ import annotations

import ctypes
from contextlib import contextmanager

_roctx = None


def _load():
    global _roctx
    if _roctx is None:
        _roctx = False
        for name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _roctx = lib
                break
            except OSError:
                continue
    return _roctx


def available() -> bool:
    return bool(_load())


@contextmanager
def range(name: str):  # noqa: A001 (mirrors the roctx name)
    lib = _load()
    if lib:
        lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        if lib:
            lib.roctxRangePop()

"""z-order de-duplication of rasterised points, on the GPU.  Mirror of salve/utils/zorder_utils.py:10-83."""

from __future__ import annotations

import ctypes

import numpy as np


def choose_elevated_repeated_vals(x: np.ndarray, y: np.ndarray, z: np.ndarray, zmin: float = -2, zmax: float = 2,
                                  num_slices: int = 4) -> np.ndarray:
    """Boolean mask of the points that win their (x, y) cell: highest occupied z slice of
    np.linspace(zmin, zmax, num_slices + 1) (half-open slices), last index within the slice; points outside
    [zmin, zmax) never win."""
    import torch

    from salve_amd import _lib

    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.SalveHipError("choose_elevated_repeated_vals runs on the HIP device only")
    dev = torch.device("cuda", torch.cuda.current_device())
    n = int(x.shape[0])
    if n == 0:
        return np.zeros(0, dtype=bool)
    w, h = int(x.max()) + 1, int(y.max()) + 1
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32)).to(dev)
    yd = torch.from_numpy(np.ascontiguousarray(y, dtype=np.int32)).to(dev)
    zd = torch.from_numpy(np.ascontiguousarray(z, dtype=np.float64)).to(dev)
    planes = torch.from_numpy(np.linspace(zmin, zmax, num_slices + 1)).to(dev)
    scratch = torch.empty(w * h, dtype=torch.int64, device=dev)
    valid = torch.empty(n, dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = lib.salve_zorder_winners(p(xd), p(yd), p(zd), n, p(planes), num_slices, w, h, p(scratch), p(valid),
                                  ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(st, "salve_zorder_winners")
    return valid.cpu().numpy().astype(bool)

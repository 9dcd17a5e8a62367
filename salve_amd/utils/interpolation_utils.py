"""Sparse-to-dense interpolation and de-hallucination mask, on the GPU.
Mirror of salve/utils/interpolation_utils.py:21-54, 74-122."""

from __future__ import annotations

import ctypes

import numpy as np

from salve_amd.common.bevparams import BEVParams

DEFAULT_KERNEL_SZ = 11
MIN_REQUIRED_POINTS_SIMPLEX = 4


def _dev():
    import torch

    from salve_amd import _lib

    if not torch.cuda.is_available():
        raise _lib.SalveHipError("salve_amd.utils.interpolation_utils runs on the HIP device only")
    return torch.device("cuda", torch.cuda.current_device())


def interp_dense_grid_from_sparse(bev_img: np.ndarray, points: np.ndarray, rgb_values: np.ndarray, grid_h: int, grid_w: int,
                                  is_semantics: bool) -> np.ndarray:
    """Linear (Delaunay) interpolation of the sparse pixels over the whole grid; `bev_img` is returned unchanged when
    there are fewer than 4 points or they all share an x or a y (:39-43).  Colours are quantised to uint8 first (in the
    renderer they are integers already); the triangulation is the canonical one of DESIGN.md section 2."""
    import torch

    from salve_amd import _lib
    from salve_amd.rasteriser import BevRasteriser

    if is_semantics:
        raise NotImplementedError("nearest-neighbour (semantic) interpolation is not part of the accelerated path")
    if points.shape[0] < MIN_REQUIRED_POINTS_SIMPLEX or np.allclose(points[:, 0], points[0, 0]) or np.allclose(points[:, 1], points[0, 1]):
        return bev_img
    dev = _dev()
    ras = BevRasteriser(dev, bev_params=BEVParams(img_h=grid_h - 1, img_w=grid_w - 1, meters_per_px=1.0))
    ras.cfg.out_flags = 3  # no flip, no mask: the plain interpolant
    xy = torch.from_numpy(np.ascontiguousarray(points[:, :2], dtype=np.int32)).to(dev)
    rgb = torch.from_numpy(np.ascontiguousarray(rgb_values).astype(np.uint8)).to(dev)
    bev = torch.empty((1, grid_h, grid_w), dtype=torch.int32, device=dev)
    ras.keys_from_pixels(xy, rgb, bev)
    ras.densify(1, bev)
    out = ras.export_u8(bev)[0].cpu().numpy()
    ras.check("interp_dense_grid_from_sparse")   # a star walk that did not close = an incomplete interpolant: never silently
    bev_img[...] = out
    return bev_img


def remove_hallucinated_content(sparse_bev_img: np.ndarray, interp_bev_img: np.ndarray, K: int = DEFAULT_KERNEL_SZ) -> np.ndarray:
    """Zero the interpolated pixels whose K x K neighbourhood holds no sparse measurement.  "Non-empty" is the uint8
    product of the three channels being non-zero -- it wraps modulo 256 exactly like the reference's (:95-98)."""
    import torch

    from salve_amd import _lib

    dev = _dev()
    lib = _lib.load()
    H, W, _ = interp_bev_img.shape
    sp = torch.from_numpy(np.ascontiguousarray(sparse_bev_img).astype(np.uint8)).to(dev)
    it = torch.from_numpy(np.ascontiguousarray(interp_bev_img).astype(np.uint8)).to(dev)
    scratch = torch.empty(H * W, dtype=torch.uint8, device=dev)
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = lib.salve_remove_hallucinated(p(sp), p(it), H, W, K, p(scratch), p(out), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(st, "salve_remove_hallucinated")
    return out.cpu().numpy()

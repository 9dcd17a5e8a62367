"""CPU restatement of the BEV texture-map rasteriser.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  numpy + oracle/csrc/delaunay_exact.c.
Every function cites the reference lines it follows (paths relative to /root/reference).
File inputs of the reference are replaced by arrays (uint16 depth, uint8 RGB) because
imageio / cv2 are not in this image and the fused GPU path never touches files.

Two densification modes (row a7):
  * "scipy": literally what the reference calls, scipy.interpolate.griddata(linear), with the
    points in the reference's order.  Pinned bit-for-bit by tests/golden (same scipy here).
  * "exact": the canonical restatement -- symbolically perturbed exact Delaunay + exact
    rational barycentric interpolation (oracle/csrc/delaunay_exact.c).  The HIP path must
    equal this mode bit for bit; it differs from "scipy" only (i) inside co-circular
    (degenerate) configurations where Qhull's choice is input-order dependent and (ii) by
    one grey level where scipy's float result lands just under an integer before the
    reference's truncating uint8 store.
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from oracle.build import build as _build_lib

_LIB = None


def _lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        path = _build_lib()
        lib = ctypes.CDLL(str(path))
        i32p = ctypes.POINTER(ctypes.c_int32)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        f64p = ctypes.POINTER(ctypes.c_double)
        lib.salve_oracle_delaunay.argtypes = [i32p, i32p, ctypes.c_int32, i32p]
        lib.salve_oracle_delaunay.restype = ctypes.c_int
        lib.salve_oracle_rasterize.argtypes = [
            i32p, i32p, u8p, i32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, u8p, f64p, u8p,
        ]
        lib.salve_oracle_rasterize.restype = None
        lib.salve_oracle_check_delaunay.argtypes = [i32p, i32p, ctypes.c_int32, i32p, ctypes.c_int32]
        lib.salve_oracle_check_delaunay.restype = ctypes.c_int
        lib.salve_oracle_tri_degenerate.argtypes = [i32p, i32p, i32p, ctypes.c_int32, u8p, ctypes.c_int32, ctypes.c_int32, u8p]
        lib.salve_oracle_tri_degenerate.restype = None
        lib.salve_oracle_rot2.argtypes = [f64p, ctypes.c_int64, f64p, f64p, f64p]
        lib.salve_oracle_rot2.restype = None
        _LIB = lib
    return _LIB


def _ptr(a: np.ndarray, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


# --------------------------------------------------------------------------- a1
def sphere_table(H: int, W: int) -> np.ndarray:
    """[H,W,3] f64 unit directions.  Follows salve/utils/hohonet_pano_utils.py:27-43."""
    v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    theta = -(u + 0.5) / W
    theta *= 2 * np.pi
    phi = (v + 0.5) / H
    phi -= 0.5
    phi *= np.pi
    r = np.cos(phi)
    return np.stack([r * np.cos(theta), r * np.sin(theta), -np.sin(phi)], -1)


# --------------------------------------------------------------------------- a2
def xyzrgb_from_arrays(
    depth_u16: np.ndarray,
    rgb_u8: np.ndarray,
    crop_z_range: Sequence[float],
    scale: float = 0.001,
    crop_ratio: float = 80 / 512,
    return_index: bool = False,
):
    """Back-project a panorama.  Follows salve/utils/bev_rendering_utils.py:367,391-413.

    depth is uint16 (the .depth.png payload); rgb is already at the working resolution
    (the cv2.resize of :375 is a pano-ingest step upstream of the hot path).
    Returns [M,6] f64 (x, y, z, r/255, g/255, b/255) and optionally the raster index of
    every kept point inside the row-cropped panorama.
    """
    H, W = rgb_u8.shape[:2]
    depth = depth_u16[..., None].astype(np.float32) * np.float32(scale)  # float32 product, :367
    xyz = depth * sphere_table(H, W)  # float32 * float64 -> float64, :392
    xyzrgb = np.concatenate([xyz, rgb_u8 / 255.0], 2)  # :394
    if crop_ratio > 0:
        crop = int(H * crop_ratio)  # :399
        xyzrgb = xyzrgb[crop:-crop]
    xyzrgb = xyzrgb.reshape(-1, 6)
    keep = np.logical_and(xyzrgb[:, 2] > crop_z_range[0], xyzrgb[:, 2] <= crop_z_range[1])  # :408-411
    if return_index:
        return xyzrgb[keep], np.nonzero(keep)[0]
    return xyzrgb[keep]


# --------------------------------------------------------------------------- a3
def rot2(xy: np.ndarray, R: np.ndarray, t: Optional[np.ndarray] = None) -> np.ndarray:
    """`xy @ R.T (+ t)` with the exact rounding of the reference's BLAS call.

    numpy evaluates `(N,2) @ (2,2)` through OpenBLAS dgemm whose FMA micro-kernel computes
    each output as fma(y, R[i,1], x*R[i,0]) (probed in this image: bit-identical on 2e5
    random points, while the un-fused form differs on ~30 % of them).  The reference's call
    sites are bev_rendering_utils.py:445-446,451 and sim2.py:157.
    """
    xy = np.ascontiguousarray(xy, dtype=np.float64)
    R64 = np.ascontiguousarray(np.asarray(R, dtype=np.float64).reshape(4))
    out = np.empty_like(xy)
    tt = None if t is None else np.ascontiguousarray(np.asarray(t, dtype=np.float64).reshape(2))
    _lib().salve_oracle_rot2(
        _ptr(xy, ctypes.c_double), ctypes.c_int64(xy.shape[0]), _ptr(R64, ctypes.c_double),
        None if tt is None else _ptr(tt, ctypes.c_double), _ptr(out, ctypes.c_double),
    )
    return out


def rotmat2d(theta_deg: float) -> np.ndarray:
    """salve/utils/rotation_utils.py:14-29 (note cos(-90 deg) = 6.1e-17, not 0)."""
    th = np.deg2rad(theta_deg)
    s, c = np.sin(th), np.cos(th)
    return np.array([[c, -s], [s, c]])


HOHO_S_ZIND_SCALE_FACTOR = 1.5  # bev_rendering_utils.py:448


def pose_pair(xyzrgb1: np.ndarray, xyzrgb2: np.ndarray, R32: np.ndarray, t32: np.ndarray):
    """In-place-equivalent of bev_rendering_utils.py:443-451.  R32, t32 are float32 (sim2.py:50-51)."""
    Rm90 = rotmat2d(-90)
    a = xyzrgb1.copy()
    b = xyzrgb2.copy()
    a[:, :2] = rot2(a[:, :2], Rm90)
    b[:, :2] = rot2(b[:, :2], Rm90)
    R32 = np.asarray(R32, dtype=np.float32)
    t_scaled = np.asarray(t32, dtype=np.float32) * HOHO_S_ZIND_SCALE_FACTOR  # float32 product, :451
    assert t_scaled.dtype == np.float32
    a[:, :2] = rot2(a[:, :2], R32.astype(np.float64), t_scaled.astype(np.float64))
    return a, b


# --------------------------------------------------------------------------- a4
@dataclass(frozen=True)
class BevGrid:
    """salve/common/bevparams.py:28-78 reduced to the numbers the rasteriser uses."""

    img_h: int = 500
    img_w: int = 500
    meters_per_px: float = 0.02

    @property
    def lims(self) -> Tuple[int, int, int, int]:
        hx = int((self.img_w / 2) * self.meters_per_px)
        hy = int((self.img_h / 2) * self.meters_per_px)
        return -hx, hx, -hy, hy  # xmin, xmax, ymin, ymax

    @property
    def scale(self) -> float:
        return float(1 / self.meters_per_px)

    @property
    def H(self) -> int:
        return self.img_h + 1

    @property
    def W(self) -> int:
        return self.img_w + 1


def bev_pixel_indices(xyz: np.ndarray, grid: BevGrid = BevGrid()):
    """prune_to_2d_bbox (:38-45) + bevimg_Sim2_world.transform_from (sim2.py:157-160, with
    R = I, t = (-xmin, -ymin) as float32, s = 1/m_per_px) + np.round (:287).
    Returns (kept mask over the input rows, img_xy int64 [M',2])."""
    xmin, xmax, ymin, ymax = grid.lims
    x, y = xyz[:, 0], xyz[:, 1]
    kept = np.logical_and.reduce([xmin <= x, x <= xmax, ymin <= y, y <= ymax])
    xy = xyz[kept, :2]
    t = np.array([-xmin, -ymin], dtype=np.float32).astype(np.float64)
    img = rot2(xy, np.eye(2), t) * grid.scale
    return kept, np.round(img).astype(np.int64)


# --------------------------------------------------------------------------- a5
def choose_elevated(x: np.ndarray, y: np.ndarray, z: np.ndarray, zmin: float = -2, zmax: float = 2,
                    num_slices: int = 4) -> np.ndarray:
    """Winner per (x, y) cell = last-indexed point of the highest occupied z slice.

    Restates salve/utils/zorder_utils.py:10-83 without the image: slices are the half-open
    intervals of np.linspace(zmin, zmax, num_slices+1) (:49,56-59); points outside
    [zmin, zmax) never win (:59); within a slice numpy's repeated-index assignment keeps the
    last point (:65); higher slices overwrite lower ones.
    """
    n = x.shape[0]
    planes = np.linspace(zmin, zmax, num_slices + 1)
    sl = np.full(n, -1, dtype=np.int64)
    for k in range(num_slices):
        sl[np.logical_and(z >= planes[k], z < planes[k + 1])] = k
    cand = np.nonzero(sl >= 0)[0]
    valid = np.zeros(n, dtype=bool)
    if cand.size == 0:
        return valid
    cell = y[cand].astype(np.int64) * (int(x.max()) + 1) + x[cand].astype(np.int64)
    order = np.lexsort((cand, sl[cand], cell))  # by cell, then slice, then index
    cs = cell[order]
    last_of_cell = np.ones(cs.shape[0], dtype=bool)
    last_of_cell[:-1] = cs[1:] != cs[:-1]
    valid[cand[order[last_of_cell]]] = True
    return valid


# --------------------------------------------------------------------------- a7
def _is_degenerate(points: np.ndarray) -> bool:
    """interpolation_utils.py:39-43,57-71: < 4 points, or all x equal, or all y equal."""
    if points.shape[0] < 4:
        return True
    return bool(np.allclose(points[:, 0], points[0, 0]) or np.allclose(points[:, 1], points[0, 1]))


def delaunay_exact(px: np.ndarray, py: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Canonical (symbolically perturbed) Delaunay triangulation of distinct lattice sites.
    Returns (order, tri): `order` sorts the sites by (y, x); tri [T,3] indexes the SORTED sites."""
    order = np.lexsort((px, py))
    sx = np.ascontiguousarray(px[order], dtype=np.int32)
    sy = np.ascontiguousarray(py[order], dtype=np.int32)
    n = sx.shape[0]
    tri = np.empty((2 * n + 8, 3), dtype=np.int32)
    nt = _lib().salve_oracle_delaunay(_ptr(sx, ctypes.c_int32), _ptr(sy, ctypes.c_int32), n, _ptr(tri, ctypes.c_int32))
    if nt < 0:
        raise MemoryError("delaunay oracle allocation failed")
    return order, tri[:nt].copy()


def interp_exact(points: np.ndarray, rgb_u8: np.ndarray, H: int, W: int):
    """Exact linear interpolation on the canonical Delaunay triangulation.
    points [K,2] int (x, y), rgb_u8 [K,3].  Returns (u8 [H,W,3], f64 [H,W,3] NaN outside hull,
    cover [H,W] bool, tri [T,3] over sites sorted by (y,x), order)."""
    out = np.zeros((H, W, 3), dtype=np.uint8)
    f = np.full((H, W, 3), np.nan)
    cover = np.zeros((H, W), dtype=np.uint8)
    if _is_degenerate(points):
        return out, f, cover.astype(bool), np.zeros((0, 3), np.int32), np.arange(points.shape[0])
    order, tri = delaunay_exact(points[:, 0], points[:, 1])
    sx = np.ascontiguousarray(points[order, 0], dtype=np.int32)
    sy = np.ascontiguousarray(points[order, 1], dtype=np.int32)
    col = np.ascontiguousarray(rgb_u8[order], dtype=np.uint8)
    tri_c = np.ascontiguousarray(tri, dtype=np.int32)
    _lib().salve_oracle_rasterize(
        _ptr(sx, ctypes.c_int32), _ptr(sy, ctypes.c_int32), _ptr(col, ctypes.c_uint8), _ptr(tri_c, ctypes.c_int32),
        tri_c.shape[0], H, W, _ptr(out, ctypes.c_uint8), _ptr(f, ctypes.c_double), _ptr(cover, ctypes.c_uint8),
    )
    return out, f, cover.astype(bool), tri, order


def check_delaunay(points_sorted_xy: np.ndarray, tri: np.ndarray) -> bool:
    sx = np.ascontiguousarray(points_sorted_xy[:, 0], dtype=np.int32)
    sy = np.ascontiguousarray(points_sorted_xy[:, 1], dtype=np.int32)
    t = np.ascontiguousarray(tri, dtype=np.int32)
    return bool(_lib().salve_oracle_check_delaunay(_ptr(sx, ctypes.c_int32), _ptr(sy, ctypes.c_int32), sx.shape[0],
                                                   _ptr(t, ctypes.c_int32), t.shape[0]))


def ccw(points_sorted_xy: np.ndarray, tri: np.ndarray) -> np.ndarray:
    """Re-orient triangles counter-clockwise."""
    p = points_sorted_xy.astype(np.int64)
    a, b, c = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
    o = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    t = tri.copy()
    t[o < 0] = t[o < 0][:, [0, 2, 1]]
    return t


def check_delaunay_windowed(points_sorted_xy: np.ndarray, tri: np.ndarray) -> bool:
    """Empty-circumcircle check of `tri` against ALL sites, exact, using a KD-tree to find candidates."""
    from scipy.spatial import cKDTree

    p = points_sorted_xy.astype(np.int64)
    tree = cKDTree(p)
    A, B, C = (p[tri[:, k]].astype(np.float64) for k in range(3))
    d = 2 * (A[:, 0] * (B[:, 1] - C[:, 1]) + B[:, 0] * (C[:, 1] - A[:, 1]) + C[:, 0] * (A[:, 1] - B[:, 1]))
    sa, sb, sc = (A ** 2).sum(1), (B ** 2).sum(1), (C ** 2).sum(1)
    ux = (sa * (B[:, 1] - C[:, 1]) + sb * (C[:, 1] - A[:, 1]) + sc * (A[:, 1] - B[:, 1])) / d
    uy = (sa * (C[:, 0] - B[:, 0]) + sb * (A[:, 0] - C[:, 0]) + sc * (B[:, 0] - A[:, 0])) / d
    r = np.hypot(ux - A[:, 0], uy - A[:, 1])
    cand = tree.query_ball_point(np.stack([ux, uy], 1), r + 1e-6)
    for k, lst in enumerate(cand):
        a, b, c = (int(v) for v in tri[k])
        if (p[b][0] - p[a][0]) * (p[c][1] - p[a][1]) - (p[b][1] - p[a][1]) * (p[c][0] - p[a][0]) <= 0:
            return False
        for q in lst:
            if q in (a, b, c):
                continue
            adx, ady = (int(v) for v in p[a] - p[q])
            bdx, bdy = (int(v) for v in p[b] - p[q])
            cdx, cdy = (int(v) for v in p[c] - p[q])
            ad, bd, cd = adx * adx + ady * ady, bdx * bdx + bdy * bdy, cdx * cdx + cdy * cdy
            det = adx * (bdy * cd - bd * cdy) - ady * (bdx * cd - bd * cdx) + ad * (bdx * cdy - bdy * cdx)
            if det > 0:
                return False
    return True


def strongly_delaunay_pixels(site_xy_sorted: np.ndarray, tri: np.ndarray, H: int, W: int) -> np.ndarray:
    """[H,W] bool: pixels covered ONLY by triangles whose circumcircle carries no fourth site.  On those
    pixels every valid Delaunay triangulation (Qhull's included) interpolates within the same triangle, so
    the exact value and scipy's float value must agree to round-off (parity tier B); elsewhere the
    reference's own output depends on Qhull's input order (tier C)."""
    sx = np.ascontiguousarray(site_xy_sorted[:, 0], dtype=np.int32)
    sy = np.ascontiguousarray(site_xy_sorted[:, 1], dtype=np.int32)
    occ = np.zeros((H, W), dtype=np.uint8)
    occ[sy, sx] = 1
    t = np.ascontiguousarray(tri, dtype=np.int32)
    flags = np.zeros(t.shape[0], dtype=np.uint8)
    _lib().salve_oracle_tri_degenerate(_ptr(sx, ctypes.c_int32), _ptr(sy, ctypes.c_int32), _ptr(t, ctypes.c_int32),
                                       t.shape[0], _ptr(occ, ctypes.c_uint8), H, W, _ptr(flags, ctypes.c_uint8))
    col = np.zeros((sx.shape[0], 3), dtype=np.uint8)
    good = np.zeros((H, W), dtype=np.uint8)
    bad = np.zeros((H, W), dtype=np.uint8)
    scratch = np.zeros((H, W, 3), dtype=np.uint8)
    for sel, cov in ((flags == 0, good), (flags == 1, bad)):
        tt = np.ascontiguousarray(t[sel])
        _lib().salve_oracle_rasterize(_ptr(sx, ctypes.c_int32), _ptr(sy, ctypes.c_int32), _ptr(col, ctypes.c_uint8),
                                      _ptr(tt, ctypes.c_int32), tt.shape[0], H, W, _ptr(scratch, ctypes.c_uint8), None,
                                      _ptr(cov, ctypes.c_uint8))
    return np.logical_and(good.astype(bool), ~bad.astype(bool))


def interp_scipy(points: np.ndarray, rgb_f64: np.ndarray, H: int, W: int):
    """The reference's own call (interpolation_utils.py:45-53).  Returns (u8 image, f64 values)."""
    import scipy.interpolate

    out = np.zeros((H, W, 3), dtype=np.uint8)
    f = np.full((H, W, 3), np.nan)
    if _is_degenerate(points):
        return out, f
    gx, gy = np.meshgrid(np.linspace(0, W - 1, W), np.linspace(0, H - 1, H))  # mesh_grid.py:24-35
    xi = np.hstack([gx.reshape(-1, 1), gy.reshape(-1, 1)])
    vals = scipy.interpolate.griddata(points=points[:, :2], values=rgb_f64, xi=xi, method="linear")
    Y = xi[:, 1].astype(np.int32)
    X = xi[:, 0].astype(np.int32)
    f[Y, X, :] = vals
    with np.errstate(invalid="ignore"):
        out[Y, X, :] = vals  # NaN -> 0 and truncation, as the reference's uint8 store does on x86
    return out, f


# --------------------------------------------------------------------------- a8
def nonempty_mask(sparse_u8: np.ndarray) -> np.ndarray:
    """interpolation_utils.py:95,98: product of the three uint8 channels, WRAPPING mod 256, > 0."""
    if sparse_u8.dtype == np.uint8:
        prod = (sparse_u8[:, :, 0].astype(np.uint32) * sparse_u8[:, :, 1] * sparse_u8[:, :, 2]) & 0xFF
    else:  # the reference's KAT feeds int64 images; no wrap there
        prod = sparse_u8[:, :, 0] * sparse_u8[:, :, 1] * sparse_u8[:, :, 2]
    return prod > 0


def box_dilate(nonempty: np.ndarray, K: int) -> np.ndarray:
    """counts > 0 of a KxK all-ones convolution with zero padding K//2 (interpolation_utils.py:101-111),
    evaluated with a summed-area table instead of F.conv2d."""
    H, W = nonempty.shape
    p = K // 2
    sat = np.zeros((H + 1, W + 1), dtype=np.int64)
    sat[1:, 1:] = np.cumsum(np.cumsum(nonempty.astype(np.int64), 0), 1)
    # window of output (i, j): rows i-p .. i-p+K-1 (even K is asymmetric exactly like conv2d)
    r0 = np.clip(np.arange(H) - p, 0, H)
    r1 = np.clip(np.arange(H) - p + K, 0, H)
    c0 = np.clip(np.arange(W) - p, 0, W)
    c1 = np.clip(np.arange(W) - p + K, 0, W)
    cnt = sat[r1][:, c1] - sat[r0][:, c1] - sat[r1][:, c0] + sat[r0][:, c0]
    return cnt > 0


def remove_hallucinated(sparse: np.ndarray, interp: np.ndarray, K: int = 11) -> np.ndarray:
    """interpolation_utils.py:74-122."""
    mask = box_dilate(nonempty_mask(sparse), K)
    return (mask[:, :, None].astype(np.float32) * interp).astype(np.uint8)


# --------------------------------------------------------------------------- a4-a8 end to end
def render_bev_image(xyzrgb: np.ndarray, grid: BevGrid = BevGrid(), mode: str = "exact") -> Optional[Dict[str, np.ndarray]]:
    """salve/utils/bev_rendering_utils.py:254-328.  Returns None iff no point falls in the window (:279).
    The dict carries every intermediate the parity tests compare."""
    xyz = xyzrgb[:, :3]
    rgb = xyzrgb[:, 3:] * 255  # :267
    kept, img_xy = bev_pixel_indices(xyz, grid)
    if img_xy.shape[0] == 0:
        return None
    rgb = rgb[kept]
    z = xyz[kept, 2]
    valid = choose_elevated(img_xy[:, 0], img_xy[:, 1], z)  # :300
    pts = img_xy[valid]
    col = rgb[valid]
    H, W = grid.H, grid.W
    sparse = np.zeros((H, W, 3), dtype=np.uint8)
    sparse[pts[:, 1], pts[:, 0]] = col  # :307-308, float64 -> uint8 truncation
    res: Dict[str, np.ndarray] = {"kept": kept, "img_xy": img_xy, "valid": valid, "sparse": sparse}
    if mode == "scipy":
        interp, f = interp_scipy(pts, col, H, W)
    elif mode == "exact":
        col_u8 = sparse[pts[:, 1], pts[:, 0]]
        interp, f, cover, tri, order = interp_exact(pts, col_u8, H, W)
        res["cover"] = cover
        res["tri"] = tri
        res["site_xy_sorted"] = pts[order]
    else:
        raise ValueError(mode)
    res["interp"] = interp
    res["interp_f64"] = f
    res["mask"] = box_dilate(nonempty_mask(sparse), 11)
    res["bev"] = np.flipud(remove_hallucinated(sparse, interp))  # :318-319
    return res


# --------------------------------------------------------------------------- a9
INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def _linear_coeffs(dst: int, src: int):
    """Per-destination source index and the two fixed-point taps of OpenCV's INTER_LINEAR
    (imgproc resize.cpp, `resize` set-up loop): fx = float((d+0.5)*scale-0.5), s = floor(fx),
    taps = saturate_cast<short>(float * 2048) (round half to even), edge clamps as OpenCV."""
    scale = float(src) / float(dst)  # scale_x = 1/inv_scale_x, both double
    d = np.arange(dst, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(fx).astype(np.int64)
    fx = (fx - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    fx[lo] = 0
    s[lo] = 0
    hi = s >= src - 1
    fx[hi] = 0
    s[hi] = src - 1
    a1 = np.rint(fx * np.float32(INTER_RESIZE_COEF_SCALE)).astype(np.int64)
    a0 = np.rint((np.float32(1.0) - fx) * np.float32(INTER_RESIZE_COEF_SCALE)).astype(np.int64)
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, a0, a1


def resize_linear_u8(img: np.ndarray, out_hw: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC, as used by
    salve/utils/transform.py:256-272.  cv2 is not importable here (parity unpinned); this is the
    published 11-bit fixed-point algorithm: horizontal pass into int32 (x2048), vertical pass
    ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2."""
    h, w = out_hw
    H, W = img.shape[:2]
    sx0, sx1, ax0, ax1 = _linear_coeffs(w, W)
    sy0, sy1, by0, by1 = _linear_coeffs(h, H)
    src = img.astype(np.int64)
    hor = src[:, sx0, :] * ax0[None, :, None] + src[:, sx1, :] * ax1[None, :, None]  # [H, w, C] x2048
    S0 = hor[sy0]
    S1 = hor[sy1]
    out = (((by0[:, None, None] * (S0 >> 4)) >> 16) + ((by1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_pano_u8(img: np.ndarray, out_hw: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(rgb, (w, h), interpolation=cv2.INTER_LINEAR) as applied to every panorama at
    salve/utils/bev_rendering_utils.py:370-375 (2048x1024 -> 1024x512).  OpenCV's resize() replaces INTER_LINEAR by its
    INTER_AREA fast path when both scale factors are exactly 2 (imgproc/src/resize.cpp: "in case of scale_x && scale_y
    is equal to 2, INTER_AREA (fast) also is equal to INTER_LINEAR"): the rounded 2x2 box mean (a+b+c+d+2)>>2.
    cv2 is not importable here: parity with OpenCV itself is unpinned; sizes already equal are returned unchanged."""
    h, w = out_hw
    H, W = img.shape[:2]
    if (H, W) == (h, w):
        return img
    if H == 2 * h and W == 2 * w:
        a = img.astype(np.int64)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    return resize_linear_u8(img, out_hw)


def imagenet_mean_std():
    """salve/utils/normalization_utils.py:13-26 (values on the 0-255 scale)."""
    mean = [item * 255 for item in [0.485, 0.456, 0.406]]
    std = [item * 255 for item in [0.229, 0.224, 0.225]]
    return mean, std


def tile_from_bev(bev_u8: np.ndarray, resize_hw=(234, 234), crop_hw=(224, 224)) -> np.ndarray:
    """Val/test transform of one tile: Resize -> centre Crop -> ToTensor -> Normalize
    (salve/train_utils.py:126-159; transform.py:256-272, 386-420, 79-85, 177-202).
    Returns float32 [3, ch, cw]."""
    r = resize_linear_u8(bev_u8, resize_hw)
    ho = int((resize_hw[0] - crop_hw[0]) / 2)
    wo = int((resize_hw[1] - crop_hw[1]) / 2)
    c = r[ho:ho + crop_hw[0], wo:wo + crop_hw[1]]
    t = c.transpose(2, 0, 1).astype(np.float32)
    mean, std = imagenet_mean_std()
    for ch in range(3):
        t[ch] = (t[ch] - np.float32(mean[ch])) / np.float32(std[ch])
    return t


# --------------------------------------------------------------------------- pair level
def floor_ceiling_z_range(surface: str):
    """bev_rendering_utils.py:560-566."""
    if surface == "floor":
        return [-float("inf"), -1.0]
    if surface == "ceiling":
        return [0.5, float("inf")]
    raise ValueError(surface)


def render_bev_pair(rgb1, depth1, rgb2, depth2, R32, t32, surface: str, mode: str = "exact", grid: BevGrid = BevGrid()):
    """bev_rendering_utils.py:417-480 on arrays.  Returns (res1, res2) dicts, or (None, None)."""
    zr = floor_ceiling_z_range(surface)
    a = xyzrgb_from_arrays(depth1, rgb1, zr)
    b = xyzrgb_from_arrays(depth2, rgb2, zr)
    a, b = pose_pair(a, b, R32, t32)
    r1 = render_bev_image(a, grid, mode)
    r2 = render_bev_image(b, grid, mode)
    if r1 is None or r2 is None:
        return None, None
    return r1, r2

"""CPU restatement of the rasterised-LAYOUT modality -- TEST INFRASTRUCTURE, see oracle/__init__.py.

Reference: salve/utils/bev_rendering_utils.py:48-251 -- `rasterize_room_layout_pair` (:48-101), `rasterize_single_layout`
(:104-156), `rasterize_polygon` (:182-192, `cv2.fillPoly`), `rasterize_polyline` / `draw_polyline_cv2` (:195-251,
`cv2.line(..., thickness, lineType=cv2.LINE_AA)`), line width `bevparams.get_line_width_by_resolution` (bevparams.py:81-99:
8 px at 0.02 m/px), colours `WDO_COLOR_DICT_CV2` (:28-31).

**Parity unpinned.**  The pixel arithmetic lives in OpenCV (`cv2`, an un-vendored dependency that is not installed in this
image; the reference's tests pin nothing of it: tests/utils/test_bev_rendering_utils.py only covers prune_to_2d_bbox).  What
is restated here, all in integer arithmetic so that the HIP kernel can be checked bit for bit against it:

* `fillPoly` with its defaults (LINE_8, shift 0), from OpenCV's published algorithm (modules/imgproc/src/drawing.cpp:
  CollectPolyEdges + FillEdgeCollection): the boundary edges are drawn with the 8-connected LineIterator (error term
  `dx - 2 dy`, a minor-axis step whenever it is negative), the interior is an even-odd scanline fill on 16.16 fixed-point edge
  crossings (slope = truncated quotient, an edge covers scanlines y0 <= y < y1, a span runs from ceil(left) to floor(right)).
* the thick anti-aliased line: OpenCV builds it from an anti-aliased convex quadrilateral plus two anti-aliased discs as end
  caps, with a fixed-point coverage table.  That table is NOT restated; the rule here is the same shape -- a capsule of
  radius thickness / 2 around the segment -- with coverage `clamp(thickness / 2 + 1/2 - distance, 0, 1)` in 1/256 steps from
  the exact (integer-square-root) distance of the pixel centre, blended as dst += (colour - dst) * coverage.  Interior and
  exterior pixels agree with any correct thick line; the one-pixel anti-aliased rim may differ from OpenCV's by a few grey levels.

Everything up to the pixel coordinates (pose, the 1.5 scale factor, bevimg_Sim2_world, np.round) is plain numpy as in the reference.
"""

from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

HOHO_S_ZIND_SCALE_FACTOR = 1.5
WDO_COLOURS = {"windows": (255, 0, 0), "doors": (0, 255, 0), "openings": (0, 0, 255)}  # bev_rendering_utils.py:28-31
WHITE = (255, 255, 255)


def isqrt64(v: int) -> int:
    """floor(sqrt(v)) for 0 <= v < 2^63, the integer form the kernel uses (Newton from a float seed, then corrected)."""
    if v <= 0:
        return 0
    r = int(np.sqrt(float(v)))
    while r * r > v:
        r -= 1
    while (r + 1) * (r + 1) <= v:
        r += 1
    return r


def to_pixels(xy_world: np.ndarray, xmin: float = -5.0, ymin: float = -5.0, scale: float = 50.0) -> np.ndarray:
    """bevimg_Sim2_world.transform_from followed by np.round (rasterize_polygon :187-188): (p @ I.T + t) * s, half to even."""
    p = np.asarray(xy_world, dtype=np.float64).reshape(-1, 2)
    t = np.array([-xmin, -ymin], dtype=np.float32).astype(np.float64)   # Sim2 stores float32 (sim2.py:50-52)
    return np.round((p @ np.eye(2).T + t) * scale).astype(np.int64)


def on_line8(x: int, y: int, x1: int, y1: int, x2: int, y2: int) -> bool:
    """Is pixel (x, y) one of the pixels OpenCV's 8-connected LineIterator visits from (x1, y1) to (x2, y2)?  Closed form of
    its error recurrence: after j steps along the major axis the line has made max(0, ceil((2 dy j - dx) / (2 dx))) minor steps."""
    dx, dy = abs(x2 - x1), abs(y2 - y1)
    sx, sy = (1 if x2 >= x1 else -1), (1 if y2 >= y1 else -1)
    if dy > dx:   # steep: the roles of x and y swap
        j = (y - y1) * sy
        if j < 0 or j > dy:
            return False
        m = 0 if j == 0 else max(0, -((-(2 * dx * j - dy)) // (2 * dy)))
        return (x - x1) * sx == m
    j = (x - x1) * sx
    if j < 0 or j > dx:
        return False
    if dx == 0:
        return y == y1
    m = 0 if j == 0 else max(0, -((-(2 * dy * j - dx)) // (2 * dx)))
    return (y - y1) * sy == m


def _on_line8_row(xs: np.ndarray, y: int, e) -> np.ndarray:
    """on_line8 for a whole row of pixels (numpy form of the same closed form)."""
    x1, y1, x2, y2 = e
    dx, dy = abs(x2 - x1), abs(y2 - y1)
    sx, sy = (1 if x2 >= x1 else -1), (1 if y2 >= y1 else -1)
    if dy > dx:
        j = (y - y1) * sy
        if j < 0 or j > dy:
            return np.zeros(xs.shape, dtype=bool)
        m = 0 if j == 0 else max(0, -((-(2 * dx * j - dy)) // (2 * dy)))
        return (xs - x1) * sx == m
    j = (xs - x1) * sx
    ok = (j >= 0) & (j <= dx)
    if dx == 0:
        return ok & (y == y1)
    m = np.where(j == 0, 0, np.maximum(0, -((-(2 * dy * j - dx)) // (2 * dx))))
    return ok & ((y - y1) * sy == m)


def fill_poly(img: np.ndarray, pts: np.ndarray, colour: Sequence[int]) -> None:
    """cv2.fillPoly(img, [pts], colour) restated (see the module docstring); pts int [K, 2] (x, y), closed or not."""
    H, W = img.shape[:2]
    pts = np.asarray(pts, dtype=np.int64)
    K = len(pts)
    edges = [(int(pts[i][0]), int(pts[i][1]), int(pts[(i + 1) % K][0]), int(pts[(i + 1) % K][1])) for i in range(K)]
    xs = np.arange(W, dtype=np.int64)
    xf = xs << 16
    for y in range(H):
        cross = []
        for x0, y0, x1, y1 in edges:
            if y0 == y1:
                continue
            if y0 > y1:
                x0, y0, x1, y1 = x1, y1, x0, y0
            if y0 <= y < y1:
                num = (x1 - x0) << 16
                slope = abs(num) // (y1 - y0) * (1 if num >= 0 else -1)      # C division truncates toward zero
                cross.append((x0 << 16) + (y - y0) * slope)
        c = np.array(cross, dtype=np.int64).reshape(-1, 1)
        inside = (((c <= xf).sum(0) & 1) | ((c < xf).sum(0) & 1)).astype(bool)
        for e in edges:
            inside |= _on_line8_row(xs, y, e)
        img[y, inside] = colour


def segment_distance_256(px: int, py: int, x1: int, y1: int, x2: int, y2: int) -> int:
    """floor(256 * distance) from the pixel centre to the segment, in integers."""
    vx, vy, wx, wy = x2 - x1, y2 - y1, px - x1, py - y1
    L2 = vx * vx + vy * vy
    dot = wx * vx + wy * vy
    if L2 == 0 or dot <= 0:
        return isqrt64((wx * wx + wy * wy) << 16)
    if dot >= L2:
        ux, uy = px - x2, py - y2
        return isqrt64((ux * ux + uy * uy) << 16)
    cr = wx * vy - wy * vx
    return isqrt64(((cr * cr) << 16) // L2)


def thick_line_aa(img: np.ndarray, x1: int, y1: int, x2: int, y2: int, colour: Sequence[int], thickness: int) -> None:
    H, W = img.shape[:2]
    reach = thickness // 2 + 2
    for y in range(max(0, min(y1, y2) - reach), min(H, max(y1, y2) + reach + 1)):
        for x in range(max(0, min(x1, x2) - reach), min(W, max(x1, x2) + reach + 1)):
            cov = min(256, max(0, thickness * 128 + 128 - segment_distance_256(x, y, x1, y1, x2, y2)))
            if cov:
                for ch in range(3):
                    d = int(img[y, x, ch])
                    img[y, x, ch] = d + (((int(colour[ch]) - d) * cov + 128) >> 8)


def rasterize_single_layout(room_vertices: np.ndarray, wdos: List[Tuple[str, np.ndarray]], img_hw: Tuple[int, int] = (501, 501),
                            thickness: int = 8) -> np.ndarray:
    """rasterize_single_layout (:104-156) with render_mask=True: filled room polygon in white, every W/D/O as a thick
    anti-aliased segment in its colour, np.flipud.  room_vertices [K, 2] and wdos [(type, [2, 2])] in metres (local frame,
    already posed); the x 1.5 factor is applied here as in the reference (:127, :149)."""
    img = np.zeros((img_hw[0], img_hw[1], 3), dtype=np.uint8)
    fill_poly(img, to_pixels(np.asarray(room_vertices, dtype=np.float64) * HOHO_S_ZIND_SCALE_FACTOR), WHITE)
    for wtype, verts in wdos:
        p = to_pixels(np.asarray(verts, dtype=np.float64) * HOHO_S_ZIND_SCALE_FACTOR)
        for k in range(len(p) - 1):
            thick_line_aa(img, int(p[k][0]), int(p[k][1]), int(p[k + 1][0]), int(p[k + 1][1]), WDO_COLOURS[wtype], thickness)
    return np.flipud(img)

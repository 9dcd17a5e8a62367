"""CPU restatement of the rasterised-LAYOUT modality -- TEST INFRASTRUCTURE, see oracle/__init__.py.

Reference: salve/utils/bev_rendering_utils.py:48-251 -- `rasterize_room_layout_pair` (:48-101), `rasterize_single_layout`
(:104-156), `rasterize_polygon` (:182-192, `cv2.fillPoly`), `rasterize_polyline` / `draw_polyline_cv2` (:195-251,
`cv2.line(..., thickness, lineType=cv2.LINE_AA)`), line width `bevparams.get_line_width_by_resolution` (bevparams.py:81-99:
8 px at 0.02 m/px), colours `WDO_COLOR_DICT_CV2` (:28-31).

**Parity unpinned.**  The pixel arithmetic lives in OpenCV (`cv2`, pinned `opencv>=4.5.0` in environment_ubuntu-latest.yml:37,
an un-vendored dependency that is not installed in this image; the reference's tests pin nothing of it:
tests/utils/test_bev_rendering_utils.py only covers prune_to_2d_bbox).  What is restated here is OpenCV 4.x's published
algorithm, modules/imgproc/src/drawing.cpp, routine for routine and in its own sequential form (loops that write pixels), so
that the HIP kernel -- which evaluates the same rules per pixel in closed form -- can be checked bit for bit against it:

* `fillPoly` with its defaults (LINE_8, shift 0), CollectPolyEdges + FillEdgeCollection: the boundary edges are drawn with the
  8-connected LineIterator (error term `dx - 2 dy`, a minor-axis step whenever it is negative), the interior is an even-odd
  scanline fill on 16.16 fixed-point edge crossings (slope = truncated quotient, an edge covers scanlines y0 <= y < y1, a
  span runs from ceil(left) to floor(right)).
* `line(..., thickness = 8, LINE_AA)` = `ThickLine` (round 3: restated in full; round 2 used a capsule-distance rule of its
  own here): the four corners p +- dp with dp = cvRound(perpendicular * thickness / 2) in 16.16 fixed point, `FillConvexPoly`
  of that quadrilateral -- which first draws its four edges with `LineAA` and then fills spans ceil(left) .. floor(right) with
  the colour --, then an end cap at either end: `EllipseEx` with axes thickness / 2, i.e. `ellipse2Poly` at 30-degree steps
  (a 12-gon from the float sine table), rounded to fixed point, again `FillConvexPoly` (13 `LineAA` edges + spans).
* `LineAA`: clipLine in fixed point, major-axis stepping with a 16.16 minor coordinate, three pixels per step weighted by
  FilterTable[dist + 32], [dist], [63 - dist] (dist = 5 fractional bits), scaled by the slope correction and the end-point
  table, blended TWICE per pixel as dst += ((colour - dst) * a + 127) >> 8 (ICV_PUT_POINT applies the update two times).
The two constant tables (SlopeCorrTable, FilterTable) and the sine table's values at multiples of 30 degrees are written out
below as published.  None of this can be pinned against cv2 here: everything stays "parity unpinned", but it is now OpenCV's
rule that is implemented and not a rule of the builder's own.

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


XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT
# modules/imgproc/src/drawing.cpp: SlopeCorrTable[33], FilterTable[64]
SLOPE_CORR_TABLE = (181, 181, 181, 182, 182, 183, 184, 185, 187, 188, 190, 192, 194, 194, 197, 198,
                    201, 203, 206, 209, 211, 214, 218, 221, 224, 227, 231, 235, 238, 242, 246, 250, 254)
FILTER_TABLE = (168, 177, 185, 194, 202, 210, 218, 224, 231, 236, 241, 246, 249, 252, 254, 254,
                254, 254, 252, 249, 246, 241, 236, 231, 224, 218, 210, 202, 194, 185, 177, 168,
                158, 149, 140, 131, 122, 114, 105, 99, 91, 86, 79, 73, 67, 62, 57, 53,
                48, 44, 40, 37, 33, 30, 28, 25, 23, 21, 19, 17, 15, 14, 12, 11)
# SinTable (float, one entry per degree) at the multiples of 30 degrees the end caps use
SIN30 = {0: 0.0, 30: 0.5, 60: 0.8660254, 90: 1.0, 120: 0.8660254, 150: 0.5, 180: 0.0, 210: -0.5, 240: -0.8660254,
         270: -1.0, 300: -0.8660254, 330: -0.5, 360: -0.0, 390: 0.5, 420: 0.8660254, 450: 1.0}


def cv_round(v: float) -> int:
    """cvRound: round half to even (lrint / _mm_cvtsd_si32 under the default rounding mode)."""
    return int(np.rint(v))


def _cdiv(a: int, b: int) -> int:
    """C integer division (truncates toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def clip_line_fixed(w: int, h: int, x1: int, y1: int, x2: int, y2: int):
    """clipLine(Size2l, Point2l&, Point2l&) of drawing.cpp on fixed-point coordinates; returns None or the clipped end points."""
    right, bottom = w - 1, h - 1
    if w <= 0 or h <= 0:
        return None
    code = lambda x, y: (x < 0) + (x > right) * 2 + (y < 0) * 4 + (y > bottom) * 8
    c1, c2 = code(x1, y1), code(x2, y2)
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))     # (int64)((double)(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (x1, y1, x2, y2) if (c1 | c2) == 0 else None


def _put_point(img: np.ndarray, x: int, y: int, colour: Sequence[int], a: int) -> None:
    """ICV_PUT_POINT of LineAA for three channels: the blend is applied twice."""
    for ch in range(3):
        c = int(img[y, x, ch])
        c += ((int(colour[ch]) - c) * a + 127) >> 8
        c += ((int(colour[ch]) - c) * a + 127) >> 8
        img[y, x, ch] = c & 255


def cv_line_aa(img: np.ndarray, pt1: Tuple[int, int], pt2: Tuple[int, int], colour: Sequence[int]) -> None:
    """LineAA(img, pt1, pt2, color): end points in 16.16 fixed point."""
    H, W = img.shape[:2]
    clipped = clip_line_fixed(W << XY_SHIFT, H << XY_SHIFT, pt1[0], pt1[1], pt2[0], pt2[1])
    if clipped is None:
        return
    x1, y1, x2, y2 = clipped
    dx, dy = x2 - x1, y2 - y1
    j = -1 if dx < 0 else 0
    ax = (dx ^ j) - j
    i = -1 if dy < 0 else 0
    ay = (dy ^ i) - i
    if ax > ay:
        dy = (dy ^ j) - j
        if j:   # the three-XOR swap under the mask j: exchange the end points
            x1, x2, y1, y2 = x2, x1, y2, y1
        y_step = _cdiv(dy << XY_SHIFT, ax | 1)
        x2 += XY_ONE
        ecount = (x2 >> XY_SHIFT) - (x1 >> XY_SHIFT)
        jj = -(x1 & (XY_ONE - 1))
        y1 += ((y_step * jj) >> XY_SHIFT) + (XY_ONE >> 1)
        slope = (y_step >> (XY_SHIFT - 5)) & 0x3F
        slope ^= 0x3F if y_step < 0 else 0
        fi = (x1 >> (XY_SHIFT - 7)) & 0x78      # 4-bit fractions of the end points
        fj = (x2 >> (XY_SHIFT - 7)) & 0x78
    else:
        dx = (dx ^ i) - i
        if i:
            x1, x2, y1, y2 = x2, x1, y2, y1
        x_step = _cdiv(dx << XY_SHIFT, ay | 1)
        y2 += XY_ONE
        ecount = (y2 >> XY_SHIFT) - (y1 >> XY_SHIFT)
        jj = -(y1 & (XY_ONE - 1))
        x1 += ((x_step * jj) >> XY_SHIFT) + (XY_ONE >> 1)
        slope = (x_step >> (XY_SHIFT - 5)) & 0x3F
        slope ^= 0x3F if x_step < 0 else 0
        fi = (y1 >> (XY_SHIFT - 7)) & 0x78
        fj = (y2 >> (XY_SHIFT - 7)) & 0x78
    slope = 0x100 if (slope & 0x20) else SLOPE_CORR_TABLE[slope]
    t0 = slope << 7
    t1 = ((0x78 - fi) | 4) * slope
    t2 = (fj | 4) * slope
    ep = [0] * 9
    ep[8] = slope
    ep[1] = ep[3] = ((((fj - fi) & 0x78) | 4) * slope >> 8) & 0x1FF
    ep[2] = (t1 >> 8) & 0x1FF
    ep[4] = ((((fj - fi) + 0x80) | 4) * slope >> 8) & 0x1FF
    ep[5] = ((t1 + t0) >> 8) & 0x1FF
    ep[6] = (t2 >> 8) & 0x1FF
    ep[7] = ((t2 + t0) >> 8) & 0x1FF
    scount = 0
    if ax > ay:
        x = x1 >> XY_SHIFT
        while ecount >= 0:
            if 0 <= x < W:
                y = (y1 >> XY_SHIFT) - 1
                e = ep[(((scount >= 2) + 1) & (scount | 2)) * 3 + (((ecount >= 2) + 1) & (ecount | 2))]
                dist = (y1 >> (XY_SHIFT - 5)) & 31
                for row, a in ((y, (e * FILTER_TABLE[dist + 32] >> 8) & 0xFF), (y + 1, (e * FILTER_TABLE[dist] >> 8) & 0xFF),
                               (y + 2, (e * FILTER_TABLE[63 - dist] >> 8) & 0xFF)):
                    if 0 <= row < H:
                        _put_point(img, x, row, colour, a)
            x += 1
            y1 += y_step
            scount += 1
            ecount -= 1
    else:
        y = y1 >> XY_SHIFT
        while ecount >= 0:
            if 0 <= y < H:
                x = (x1 >> XY_SHIFT) - 1
                e = ep[(((scount >= 2) + 1) & (scount | 2)) * 3 + (((ecount >= 2) + 1) & (ecount | 2))]
                dist = (x1 >> (XY_SHIFT - 5)) & 31
                for col, a in ((x, (e * FILTER_TABLE[dist + 32] >> 8) & 0xFF), (x + 1, (e * FILTER_TABLE[dist] >> 8) & 0xFF),
                               (x + 2, (e * FILTER_TABLE[63 - dist] >> 8) & 0xFF)):
                    if 0 <= col < W:
                        _put_point(img, col, y, colour, a)
            y += 1
            x1 += x_step
            scount += 1
            ecount -= 1


def cv_fill_convex_poly_aa(img: np.ndarray, v: Sequence[Tuple[int, int]], colour: Sequence[int]) -> None:
    """FillConvexPoly(img, v, npts, color, LINE_AA, XY_SHIFT): vertices in 16.16 fixed point.  The edges are drawn with LineAA
    first, then the spans ceil(left) .. floor(right) of every scanline are filled with the colour."""
    H, W = img.shape[:2]
    npts = len(v)
    shift = XY_SHIFT
    delta = 1 << shift >> 1
    delta1, delta2 = XY_ONE - 1, 0
    p0 = v[npts - 1]
    xmin = xmax = v[0][0]
    ymin = ymax = v[0][1]
    imin = 0
    for i in range(npts):
        p = v[i]
        if p[1] < ymin:
            ymin, imin = p[1], i
        ymax = max(ymax, p[1])
        xmax = max(xmax, p[0])
        xmin = min(xmin, p[0])
        cv_line_aa(img, p0, p, colour)
        p0 = p
    xmin = (xmin + delta) >> shift
    xmax = (xmax + delta) >> shift
    ymin = (ymin + delta) >> shift
    ymax = (ymax + delta) >> shift
    if npts < 3 or xmax < 0 or ymax < 0 or xmin >= W or ymin >= H:
        return
    ymax = min(ymax, H - 1)
    edge = [dict(idx=imin, di=1, x=-XY_ONE, dx=0, ye=ymin), dict(idx=imin, di=npts - 1, x=-XY_ONE, dx=0, ye=ymin)]
    edges = npts
    y = ymin
    while True:
        if y < ymax or y == ymin:
            for e in edge:
                if y >= e["ye"]:
                    idx0, di = e["idx"], e["di"]
                    idx = idx0 + di
                    if idx >= npts:
                        idx -= npts
                    while True:
                        edges -= 1
                        if edges < 0:      # `for (; edges-- > 0; )` left with edges == -1
                            break
                        ty = (v[idx][1] + delta) >> shift
                        if ty > y:
                            xs, xe = v[idx0][0], v[idx][0]
                            e["ye"] = ty
                            e["dx"] = _cdiv((xe - xs) * 2 + (ty - y), 2 * (ty - y))
                            e["x"] = xs
                            e["idx"] = idx
                            break
                        idx0 = idx
                        idx += di
                        if idx >= npts:
                            idx -= npts
        if edges < 0:
            break
        if y >= 0:
            left, right = (1, 0) if edge[0]["x"] > edge[1]["x"] else (0, 1)
            xx1 = (edge[left]["x"] + delta1) >> XY_SHIFT
            xx2 = (edge[right]["x"] + delta2) >> XY_SHIFT
            if xx2 >= 0 and xx1 < W:
                xx1 = max(xx1, 0)
                xx2 = min(xx2, W - 1)
                if xx2 >= xx1:
                    img[y, xx1:xx2 + 1] = colour
        edge[0]["x"] += edge[0]["dx"]
        edge[1]["x"] += edge[1]["dx"]
        y += 1
        if y > ymax:
            break


def cv_ellipse_poly(cx: int, cy: int, axis: int) -> List[Tuple[int, int]]:
    """EllipseEx's polygon for a full ellipse with both axes `axis` (16.16 fixed point) around (cx, cy): ellipse2Poly at the step
    EllipseEx chooses from the axis length, each point rounded to fixed point the way EllipseEx does, consecutive duplicates
    dropped."""
    d = (axis + (XY_ONE >> 1)) >> XY_SHIFT
    delta = 90 if d < 3 else (30 if d < 10 else (18 if d < 15 else 5))
    if delta not in (30, 90):   # radii below 10 px: the reference draws 8-pixel and 2-pixel lines only (bevparams.py:81-99, :128-136)
        raise NotImplementedError("end caps of radius >= 10 px need the sine table at 18- / 5-degree steps, which is not written out here")
    alpha, beta = np.float32(SIN30[450]), np.float32(SIN30[0])     # sincos(0): cos, sin
    out: List[Tuple[int, int]] = []
    prev = None
    i = 0
    while i < 360 + delta:
        ang = min(i, 360)
        x = float(axis) * float(np.float32(SIN30[450 - ang]))
        y = float(axis) * float(np.float32(SIN30[ang]))
        px = float(cx) + x * float(alpha) - y * float(beta)
        py = float(cy) + x * float(beta) + y * float(alpha)
        qx = cv_round(px / XY_ONE) << XY_SHIFT
        qy = cv_round(py / XY_ONE) << XY_SHIFT
        qx += cv_round(px - qx)
        qy += cv_round(py - qy)
        if (qx, qy) != prev:
            out.append((qx, qy))
            prev = (qx, qy)
        i += delta
    if len(out) <= 1:
        out = [(cx, cy), (cx, cy)]
    return out


def thick_line_geometry(x1: int, y1: int, x2: int, y2: int, thickness: int):
    """ThickLine's geometry for integer pixel end points (shift 0): the quadrilateral (or None for a zero-length segment) and
    the half-thickness in fixed point (the end caps' axes), all in 16.16 fixed point."""
    p0x, p0y, p1x, p1y = x1 << XY_SHIFT, y1 << XY_SHIFT, x2 << XY_SHIFT, y2 << XY_SHIFT
    inv = 1.0 / XY_ONE
    dx, dy = (p0x - p1x) * inv, (p1y - p0y) * inv
    r = dx * dx + dy * dy
    odd = thickness & 1
    th = thickness << (XY_SHIFT - 1)
    quad = None
    if abs(r) > np.finfo(np.float64).eps:
        r = (th + odd * XY_ONE * 0.5) / np.sqrt(r)
        dpx, dpy = cv_round(dy * r), cv_round(dx * r)
        quad = [(p0x + dpx, p0y + dpy), (p0x - dpx, p0y - dpy), (p1x - dpx, p1y - dpy), (p1x + dpx, p1y + dpy)]
    return quad, th, (p0x, p0y), (p1x, p1y)


def cv_thick_line_aa(img: np.ndarray, x1: int, y1: int, x2: int, y2: int, colour: Sequence[int], thickness: int) -> None:
    """cv2.line(img, (x1, y1), (x2, y2), colour, thickness, lineType=cv2.LINE_AA) for thickness > 1: ThickLine with flags 3."""
    if thickness <= 1:
        cv_line_aa(img, (x1 << XY_SHIFT, y1 << XY_SHIFT), (x2 << XY_SHIFT, y2 << XY_SHIFT), colour)
        return
    quad, th, p0, p1 = thick_line_geometry(x1, y1, x2, y2, thickness)
    if quad is not None:
        cv_fill_convex_poly_aa(img, quad, colour)
    for c in (p0, p1):
        cv_fill_convex_poly_aa(img, cv_ellipse_poly(c[0], c[1], th), colour)


def rasterize_single_layout(room_vertices: np.ndarray, wdos: List[Tuple[str, np.ndarray]], img_hw: Tuple[int, int] = (501, 501),
                            thickness: int = 8, render_mask: bool = True) -> np.ndarray:
    """rasterize_single_layout (:104-156): the room polygon filled in white (render_mask=True) or drawn as a polyline of a third
    of the W/D/O width (:128-136), every W/D/O as a thick anti-aliased segment in its colour, np.flipud.  room_vertices [K, 2]
    and wdos [(type, [2, 2])] in metres (local frame, already posed); the x 1.5 factor is applied here as in the reference
    (:127, :149)."""
    img = np.zeros((img_hw[0], img_hw[1], 3), dtype=np.uint8)
    room_px = to_pixels(np.asarray(room_vertices, dtype=np.float64) * HOHO_S_ZIND_SCALE_FACTOR)
    if render_mask:
        fill_poly(img, room_px, WHITE)
    else:
        for k in range(len(room_px) - 1):
            cv_thick_line_aa(img, int(room_px[k][0]), int(room_px[k][1]), int(room_px[k + 1][0]), int(room_px[k + 1][1]), WHITE, int(thickness / 3))
    for wtype, verts in wdos:
        p = to_pixels(np.asarray(verts, dtype=np.float64) * HOHO_S_ZIND_SCALE_FACTOR)
        for k in range(len(p) - 1):
            cv_thick_line_aa(img, int(p[k][0]), int(p[k][1]), int(p[k + 1][0]), int(p[k + 1][1]), WDO_COLOURS[wtype], thickness)
    return np.flipud(img)

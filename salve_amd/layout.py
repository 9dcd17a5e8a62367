"""Host side of the rasterised-LAYOUT modality: room polygons and W/D/O segments of many
(panorama, pose) pairs packed into flat tables, ONE launch of salve_layout_rasterise for all of them.

Reference: salve/utils/bev_rendering_utils.py:48-251.  Everything up to the integer pixel coordinates is the reference's numpy
(pose `i2Ti1.transform_from`, the x 1.5 HoHoNet -> ZInD factor :127/:149, `bevimg_Sim2_world.transform_from`, `np.round`
:187-188); the pixel arithmetic runs in salve_amd/csrc/layout.hip.  There is no CPU renderer here.
"""

from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from salve_amd import _lib, status
from salve_amd.common.bevparams import DEFAULT_METERS_PER_PX, BEVParams, get_line_width_by_resolution

HOHO_S_ZIND_SCALE_FACTOR = 1.5
RED, GREEN, BLUE, WHITE = (255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 255)
WDO_COLOR_DICT_CV2 = {"windows": RED, "doors": GREEN, "openings": BLUE}   # bev_rendering_utils.py:28-31

# one layout image: (room vertices [K, 2] in metres, already posed; [(W/D/O type, vertices [2, 2] in metres, already posed)])
LayoutSpec = Tuple[np.ndarray, Sequence[Tuple[str, np.ndarray]]]


MAX_LAYOUTS_PER_LAUNCH = 65535   # include/salve_hip.h: salve_layout_rasterise


def world_to_pixels(bev_params: BEVParams, xy: np.ndarray) -> np.ndarray:
    """rasterize_polygon / rasterize_polyline :187-188, :214-215."""
    return np.round(bev_params.bevimg_Sim2_world.transform_from(np.asarray(xy, dtype=np.float64).reshape(-1, 2))).astype(np.int64)


def rasterise_layouts(specs: Sequence[LayoutSpec], device, bev_params: Optional[BEVParams] = None, render_mask: bool = True) -> torch.Tensor:
    """-> int32 [n, H + 1, W + 1] device tensor holding 0x00BBGGRR, flipped vertically like the texture maps (the format
    BevRasteriser.tiles / export_u8 take).  render_mask=False (a thin contour instead of the filled room, :128-136) draws the
    room boundary as a polyline of a third of the W/D/O width."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.SalveHipError("layout rasterisation needs a HIP device ('cuda:N'); there is no CPU path")
    lib = _lib.load()
    bp = bev_params or BEVParams()
    H, W = bp.img_h + 1, bp.img_w + 1
    width = get_line_width_by_resolution(DEFAULT_METERS_PER_PX)
    n = len(specs)
    rec = np.zeros(n, dtype=_lib.LAYOUT_DTYPE)
    polys: List[np.ndarray] = []
    segs: List[Tuple[int, ...]] = []
    for i, (room, wdos) in enumerate(specs):
        room_px = world_to_pixels(bp, np.asarray(room, dtype=np.float64) * HOHO_S_ZIND_SCALE_FACTOR)
        rec[i]["poly_off"], rec[i]["seg_off"] = sum(len(p) for p in polys), len(segs)
        if render_mask:
            polys.append(room_px)
            rec[i]["n_poly"] = len(room_px)
        else:
            col = WHITE[0] | (WHITE[1] << 8) | (WHITE[2] << 16)
            for k in range(len(room_px) - 1):
                segs.append((*room_px[k], *room_px[k + 1], col, int(width / 3), 0, 0))
        for wtype, verts in wdos:
            c = WDO_COLOR_DICT_CV2[wtype]
            px = world_to_pixels(bp, np.asarray(verts, dtype=np.float64) * HOHO_S_ZIND_SCALE_FACTOR)
            for k in range(len(px) - 1):
                segs.append((*px[k], *px[k + 1], c[0] | (c[1] << 8) | (c[2] << 16), width, 0, 0))
        rec[i]["n_seg"] = len(segs) - rec[i]["seg_off"]
    poly_np = np.concatenate(polys).astype(np.int32) if polys else np.zeros((1, 2), np.int32)
    seg_np = np.array(segs, dtype=np.int64).astype(np.int32).reshape(-1, 8) if segs else np.zeros((1, 8), np.int32)
    # The kernel clips geometrically (OpenCV's clipLine in 64-bit fixed point; the polygon rules in 64-bit integers): coordinates
    # are NOT clamped per axis -- that would change the slopes of edges that cross the image.  Only absurd values are refused.
    if segs and int(seg_np[:, 5].max()) >= 19:
        raise ValueError("lines of 19 pixels and more need OpenCV's end caps at 18- / 5-degree steps, which the kernel does not draw "
                         "(the reference draws 8- and 2-pixel lines: bevparams.get_line_width_by_resolution(0.02))")
    if max(int(np.abs(poly_np).max(initial=0)), int(np.abs(seg_np[:, :4]).max(initial=0))) > (1 << 24):
        raise ValueError("layout geometry more than 2^24 pixels away from the image: not a room layout")
    d_rec = torch.from_numpy(rec.view(np.uint8)).to(device)
    d_poly = torch.from_numpy(np.ascontiguousarray(poly_np)).to(device)
    d_seg = torch.from_numpy(np.ascontiguousarray(seg_np)).to(device)
    out = torch.empty((n, H, W), dtype=torch.int32, device=device)
    rec_bytes = _lib.LAYOUT_DTYPE.itemsize
    with torch.cuda.device(device):
        # salve_layout_rasterise takes at most 65535 images per call (one grid dimension): a floor's hypotheses go in chunks;
        # the records carry absolute offsets into the shared vertex / segment arrays, so a chunk is a slice of the record table
        for lo in range(0, n, MAX_LAYOUTS_PER_LAUNCH):
            m = min(MAX_LAYOUTS_PER_LAUNCH, n - lo)
            st = lib.salve_layout_rasterise(ctypes.c_void_p(d_rec.data_ptr() + lo * rec_bytes), m, ctypes.c_void_p(d_poly.data_ptr()),
                                            ctypes.c_void_p(d_seg.data_ptr()), H, W, ctypes.c_void_p(out[lo:].data_ptr()), status.ptr(device),
                                            ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream))
            _lib.check(st, "salve_layout_rasterise")
    return out


def layout_pair_specs(i2Ti1, floor_pose_graph, i1: int, i2: int) -> Tuple[LayoutSpec, LayoutSpec]:
    """The two layouts of rasterize_room_layout_pair (:48-101): panorama i1's room and W/D/Os moved into i2's frame by i2Ti1
    (:82, :90), panorama i2's as they are (:96).  `floor_pose_graph.nodes[i]` must offer `room_vertices_local_2d` [K, 2] and
    `doors`, `windows`, `openings`: lists of objects with `.type` and `.vertices_local_2d` [2, 2] (salve/common/wdo.py:47-50)."""
    n1, n2 = floor_pose_graph.nodes[i1], floor_pose_graph.nodes[i2]
    close = lambda v: np.vstack([np.asarray(v, dtype=np.float64), np.asarray(v, dtype=np.float64)[0].reshape(-1, 2)])  # :76-77
    room1 = i2Ti1.transform_from(close(n1.room_vertices_local_2d))
    room2 = close(n2.room_vertices_local_2d)
    wdos1 = [(w.type, i2Ti1.transform_from(np.asarray(w.vertices_local_2d, dtype=np.float64))) for w in list(n1.doors) + list(n1.windows) + list(n1.openings)]
    wdos2 = [(w.type, np.asarray(w.vertices_local_2d, dtype=np.float64)) for w in list(n2.doors) + list(n2.windows) + list(n2.openings)]
    return (room1, wdos1), (room2, wdos2)

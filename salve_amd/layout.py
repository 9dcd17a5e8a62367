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


class PackedLayouts:
    """Flat device tables of n layout images (include/salve_hip.h: salve_layout_t records + shared vertex / segment arrays): what
    `salve_layout_rasterise` reads.  Packed once on the host (`pack_layouts`), rasterised in any number of launches over slices of
    the record table (the records carry absolute offsets into the shared arrays)."""

    def __init__(self, rec: np.ndarray, poly: np.ndarray, seg: np.ndarray, hw: Tuple[int, int], device: torch.device) -> None:
        self.n = len(rec)
        self.hw = hw
        self.device = device
        self.rec = torch.from_numpy(rec.view(np.uint8)).to(device)
        self.poly = torch.from_numpy(np.ascontiguousarray(poly)).to(device)
        self.seg = torch.from_numpy(np.ascontiguousarray(seg)).to(device)

    def rasterise(self, lo: int, n: int, out: torch.Tensor) -> torch.Tensor:
        """Images [lo, lo + n) -> out (int32 [>= n, H, W], 0x00BBGGRR, flipped vertically like the texture maps) on the current stream."""
        assert 0 <= lo and lo + n <= self.n and out.is_contiguous() and out.numel() >= n * self.hw[0] * self.hw[1]
        lib = _lib.load()
        rec_bytes = _lib.LAYOUT_DTYPE.itemsize
        H, W = self.hw
        with torch.cuda.device(self.device):
            # salve_layout_rasterise takes at most 65535 images per call (one grid dimension)
            for a in range(lo, lo + n, MAX_LAYOUTS_PER_LAUNCH):
                m = min(MAX_LAYOUTS_PER_LAUNCH, lo + n - a)
                st = lib.salve_layout_rasterise(ctypes.c_void_p(self.rec.data_ptr() + a * rec_bytes), m, ctypes.c_void_p(self.poly.data_ptr()),
                                                ctypes.c_void_p(self.seg.data_ptr()), H, W, ctypes.c_void_p(out[a - lo:].data_ptr()), status.ptr(self.device),
                                                ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
                _lib.check(st, "salve_layout_rasterise")
        return out


def pack_layouts(specs: Sequence[LayoutSpec], device, bev_params: Optional[BEVParams] = None, render_mask: bool = True) -> PackedLayouts:
    """Host side of `rasterise_layouts`: metres -> integer pixels exactly as the reference does (x 1.5, `bevimg_Sim2_world`,
    `np.round`: bev_rendering_utils.py:127, 149, 187-188, 214-215), packed into the kernel's tables and uploaded."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.SalveHipError("layout rasterisation needs a HIP device ('cuda:N'); there is no CPU path")
    bp = bev_params or BEVParams()
    H, W = bp.img_h + 1, bp.img_w + 1
    width = get_line_width_by_resolution(DEFAULT_METERS_PER_PX)
    n = len(specs)
    rec = np.zeros(n, dtype=_lib.LAYOUT_DTYPE)
    polys: List[np.ndarray] = []
    segs: List[Tuple[int, ...]] = []
    n_poly = 0
    for i, (room, wdos) in enumerate(specs):
        room_px = world_to_pixels(bp, np.asarray(room, dtype=np.float64) * HOHO_S_ZIND_SCALE_FACTOR)
        rec[i]["poly_off"], rec[i]["seg_off"] = n_poly, len(segs)
        if render_mask:
            polys.append(room_px)
            n_poly += len(room_px)
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
    return PackedLayouts(rec, poly_np, seg_np, (H, W), device)


def rasterise_layouts(specs: Sequence[LayoutSpec], device, bev_params: Optional[BEVParams] = None, render_mask: bool = True) -> torch.Tensor:
    """-> int32 [n, H + 1, W + 1] device tensor holding 0x00BBGGRR, flipped vertically like the texture maps (the format
    BevRasteriser.tiles / export_u8 take).  render_mask=False (a thin contour instead of the filled room, :128-136) draws the
    room boundary as a polyline of a third of the W/D/O width."""
    packed = pack_layouts(specs, device, bev_params, render_mask)
    out = torch.empty((packed.n, *packed.hw), dtype=torch.int32, device=packed.device)
    return packed.rasterise(0, packed.n, out) if packed.n else out


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


class FusedLayouts:
    """The layout images the fused render -> verify pipeline needs for a hypothesis table (pipeline.RenderVerifyPipeline.prepare):
    `posed[j]` = panorama i1[j]'s room and W/D/Os under hypothesis j's i2Ti1 (rasterize_room_layout_pair :82, :90), `identity[p]` =
    panorama (store index) p's own layout (:96), the same for every hypothesis that names p as its second panorama."""

    def __init__(self, posed: Sequence[LayoutSpec], identity: Sequence[LayoutSpec]) -> None:
        self.posed, self.identity = list(posed), list(identity)

    @classmethod
    def from_pose_graph(cls, table, pano_ids: Sequence[int], floor_pose_graph, scales: Optional[Sequence[float]] = None) -> "FusedLayouts":
        """table: HypothesisTable whose i1 / i2 index `pano_ids` (ingest.PanoStore order); scales: the hypotheses' Sim(2) scales
        (default 1: alignment hypotheses are SE(2), export_alignment_hypotheses.py)."""
        from salve_amd.common.sim2 import Sim2

        posed = []
        for j in range(len(table)):
            S = Sim2(table.R[j], table.t[j], 1.0 if scales is None else float(scales[j]))
            posed.append(layout_pair_specs(S, floor_pose_graph, int(pano_ids[int(table.i1[j])]), int(pano_ids[int(table.i2[j])]))[0])
        close = lambda v: np.vstack([np.asarray(v, dtype=np.float64), np.asarray(v, dtype=np.float64)[0].reshape(-1, 2)])
        identity = []
        for pid in pano_ids:
            nd = floor_pose_graph.nodes.get(int(pid)) if hasattr(floor_pose_graph.nodes, "get") else floor_pose_graph.nodes[int(pid)]
            if nd is None:    # a panorama no hypothesis names as its second one may be missing from the graph: an empty image
                identity.append((np.zeros((0, 2)), []))
                continue
            identity.append((close(nd.room_vertices_local_2d),
                             [(w.type, np.asarray(w.vertices_local_2d, dtype=np.float64)) for w in list(nd.doors) + list(nd.windows) + list(nd.openings)]))
        return cls(posed, identity)

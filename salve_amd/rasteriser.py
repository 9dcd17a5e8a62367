"""Host side of the HIP BEV rasteriser: buffers, tables and launches (no arithmetic on the hot path).

PyTorch is used for device memory and streams only; every computation below the table set-up runs in
salve_amd/csrc/bev_render.hip through the C ABI (include/salve_hip.h).
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from salve_amd import _lib, status
from salve_amd.common.bevparams import BEVParams
from salve_amd.utils import rotation_utils
from salve_amd.utils.hohonet_pano_utils import get_sphere_factors
from salve_amd.utils.normalization_utils import get_imagenet_mean_std

HOHO_S_ZIND_SCALE_FACTOR = 1.5  # applied inside the kernel (reference bev_rendering_utils.py:448-451)
SURFACES = {"floor": 0, "ceiling": 1}
# crop_z_range per surface (reference bev_rendering_utils.py:560-566): lo < z <= hi
Z_RANGES = {"floor": (-float("inf"), -1.0), "ceiling": (0.5, float("inf"))}


def linear_resize_taps(dst: int, src: int) -> np.ndarray:
    """int32 [dst, 4] = (src0, src1, w0, w1): the two source indices and 11-bit fixed-point weights OpenCV's
    uint8 INTER_LINEAR uses for every destination index (float32 source coordinate, weights rounded half to even,
    edge clamps), as called by the reference's Resize transforms (salve/utils/transform.py:256-272)."""
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    below, above = s < 0, s >= src - 1
    f[below] = 0
    s[below] = 0
    f[above] = 0
    s[above] = src - 1
    w1 = np.rint(f * np.float32(2048)).astype(np.int64)
    w0 = np.rint((np.float32(1.0) - f) * np.float32(2048)).astype(np.int64)
    return np.stack([s, np.minimum(s + 1, src - 1), w0, w1], -1).astype(np.int32)


def normalisation_lut() -> np.ndarray:
    """float32 [3, 256]: (v - mean_c) / std_c in float32, i.e. what ToTensor + Normalize produce for a uint8 value
    (salve/utils/transform.py:79-85, 177-202; constants normalization_utils.py:13-26)."""
    mean, std = get_imagenet_mean_std()
    v = np.arange(256, dtype=np.float32)
    return np.stack([(v - np.float32(m)) / np.float32(s) for m, s in zip(mean, std)]).astype(np.float32)


def pack_hypotheses(pano_idx, surface, R, t, apply_pose) -> np.ndarray:
    """Structured array of salve_bev_hyp_t rows."""
    n = len(pano_idx)
    h = np.zeros(n, dtype=_lib.HYP_DTYPE)
    h["pano_idx"] = np.asarray(pano_idx, dtype=np.int32)
    h["surface"] = np.asarray(surface, dtype=np.int32)
    h["R"] = np.asarray(R, dtype=np.float32).reshape(n, 4)
    h["t"] = np.asarray(t, dtype=np.float32).reshape(n, 2)
    h["apply_pose"] = np.asarray(apply_pose, dtype=np.int32)
    return h


@dataclass
class RenderDebug:
    img_xy: Optional[torch.Tensor] = None  # int16 [n, npts, 2]
    keys: Optional[torch.Tensor] = None  # int64 [n, H*W]
    mask: Optional[torch.Tensor] = None  # uint8 [n, H, W]
    stats: Optional[torch.Tensor] = None  # int32 [n, 8]
    in_window: Optional[torch.Tensor] = None  # int32 [n]


class BevRasteriser:
    """Owns the device-side tables and workspace of the rasteriser for one GPU."""

    def __init__(self, device: torch.device, pano_hw: Tuple[int, int] = (512, 1024), bev_params: Optional[BEVParams] = None,
                 crop_ratio: float = 80 / 512, depth_scale: float = 0.001, mask_k: int = 11,
                 resize: int = 234, crop: int = 224) -> None:
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SalveHipError("BevRasteriser needs a HIP device ('cuda:N'); there is no CPU path")
        bp = bev_params or BEVParams()
        H, W = pano_hw
        self.pano_hw = (H, W)
        self.bev_hw = (bp.img_h + 1, bp.img_w + 1)
        self.resize, self.crop = resize, crop
        cfg = _lib.BevConfig()
        cfg.pano_h, cfg.pano_w = H, W
        cfg.crop_rows = int(H * crop_ratio) if crop_ratio > 0 else 0
        cfg.bev_h, cfg.bev_w = self.bev_hw
        cfg.mask_k = mask_k
        cfg.depth_scale = np.float32(depth_scale)
        cfg.win_xmin, cfg.win_xmax = bp.xlims
        cfg.win_ymin, cfg.win_ymax = bp.ylims
        S = bp.bevimg_Sim2_world
        cfg.img_tx, cfg.img_ty = float(S.translation[0]), float(S.translation[1])
        cfg.img_scale = float(S.scale)
        Rm = rotation_utils.rotmat2d(-90)
        for i in range(4):
            cfg.rot_pre[i] = float(Rm.reshape(4)[i])
        cfg.z_lo[0], cfg.z_hi[0] = Z_RANGES["floor"]
        cfg.z_lo[1], cfg.z_hi[1] = Z_RANGES["ceiling"]
        cfg.z_min, cfg.n_slices = -2.0, 4
        self.cfg = cfg
        self.npts = (H - 2 * cfg.crop_rows) * W
        r, zdir, ct, st = get_sphere_factors(H, W)
        self.sphere = torch.from_numpy(np.concatenate([r, zdir, ct, st])).to(self.device)
        self.coef_y = torch.from_numpy(linear_resize_taps(resize, self.bev_hw[0])).to(self.device)
        self.coef_x = torch.from_numpy(linear_resize_taps(resize, self.bev_hw[1])).to(self.device)
        self.lut = torch.from_numpy(normalisation_lut()).to(self.device)
        self._ws_slots = {}   # workspaces by slot: a caller that keeps two batches in flight alternates `ws_slot`
        self.ws_slot = 0

    # ------------------------------------------------------------------ helpers
    def _stream(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, n: int) -> torch.Tensor:
        need = self.lib.salve_bev_workspace_bytes(ctypes.byref(self.cfg), n)
        if need == 0:
            _lib.check(-1, "salve_bev_workspace_bytes")
        ws = self._ws_slots.get(self.ws_slot)
        if ws is None or ws.numel() < need:
            if ws is not None:
                # growing a slot: launches issued earlier -- possibly on OTHER streams (the pipeline runs scatter and densify on
                # their own) -- may still read the old buffer, and the caching allocator hands its memory out again as soon as
                # the last reference goes, ordered against the CURRENT stream only.  Growth is rare (a larger batch than any
                # before): wait for the device.
                torch.cuda.synchronize(self.device)
            ws = self._ws_slots[self.ws_slot] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def pano_index(self, pano_depth: torch.Tensor) -> torch.Tensor:
        """The pose-independent panorama index of these depth maps (include/salve_hip.h: salve_bev_pano_index_build), built
        on first use on the current stream and kept with the tensor OBJECT: it lives as long as the tensor does, a slice or a
        copy builds its own.  The key includes the tensor's VERSION counter (torch bumps it on every in-place write: `copy_`,
        `[...] =`, `add_` ...), so depth maps overwritten in place get a fresh index instead of silently rendering with the stale
        one; only writes torch cannot see (a kernel given the raw pointer) need `drop_pano_index`."""
        idx = getattr(pano_depth, "_salve_pano_index", None)
        P = int(pano_depth.shape[0])
        if idx is not None and idx[1] == (pano_depth.data_ptr(), P, pano_depth._version):
            return idx[0]
        assert pano_depth.is_contiguous() and tuple(pano_depth.shape[1:]) == self.pano_hw and pano_depth.element_size() == 2
        if idx is not None:
            # REPLACING an index (the depth maps were written in place): launches issued earlier -- possibly on other streams, the pipeline runs its
            # scatter on one of its own -- may still read the old buffer, and the caching allocator hands its memory out again as soon as the last
            # reference goes, ordered against the current stream only.  Rare (as rare as overwriting depth maps): wait for the device, as the
            # workspace-growth path does.
            torch.cuda.synchronize(self.device)
        nbytes = self.lib.salve_bev_pano_index_bytes(ctypes.byref(self.cfg), P)
        if nbytes == 0:
            _lib.check(-1, "salve_bev_pano_index_bytes")
        buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            st = self.lib.salve_bev_pano_index_build(ctypes.byref(self.cfg), ctypes.c_void_p(pano_depth.data_ptr()), P,
                                                     ctypes.c_void_p(self.sphere.data_ptr()), ctypes.c_void_p(buf.data_ptr()), nbytes, self._stream())
        _lib.check(st, "salve_bev_pano_index_build")
        pano_depth._salve_pano_index = (buf, (pano_depth.data_ptr(), P, pano_depth._version))
        return buf

    @staticmethod
    def drop_pano_index(pano_depth: torch.Tensor) -> None:
        if hasattr(pano_depth, "_salve_pano_index"):
            del pano_depth._salve_pano_index

    def upload_panos(self, rgb: np.ndarray, depth: np.ndarray) -> Tuple[torch.Tensor, torch.Tensor]:
        """rgb uint8 [P,H,W,3], depth uint16 [P,H,W] (host) -> device tensors (depth carried as int16 bits)."""
        assert rgb.dtype == np.uint8 and depth.dtype == np.uint16
        assert rgb.shape[1:3] == self.pano_hw and depth.shape[1:] == self.pano_hw
        d = torch.from_numpy(np.ascontiguousarray(depth).view(np.int16)).to(self.device)
        return torch.from_numpy(np.ascontiguousarray(rgb)).to(self.device), d

    def upload_hypotheses(self, hyps: np.ndarray) -> torch.Tensor:
        assert hyps.dtype == _lib.HYP_DTYPE
        return torch.from_numpy(np.ascontiguousarray(hyps).view(np.uint8)).to(self.device)

    # ------------------------------------------------------------------ launches
    def render(self, pano_rgb: torch.Tensor, pano_depth: torch.Tensor, hyps_dev: torch.Tensor, n: int,
               out_bev: Optional[torch.Tensor] = None, debug: bool = False) -> Tuple[torch.Tensor, RenderDebug]:
        """Render n BEV images.  Returns (int32 [n, H, W] holding 0x00BBGGRR, debug buffers)."""
        Hb, Wb = self.bev_hw
        if out_bev is None:
            out_bev = torch.empty((n, Hb, Wb), dtype=torch.int32, device=self.device)
        assert out_bev.is_contiguous() and out_bev.numel() >= n * Hb * Wb
        dbg = RenderDebug()
        if debug:
            dbg.img_xy = torch.empty((n, self.npts, 2), dtype=torch.int16, device=self.device)
            dbg.keys = torch.empty((n, Hb * Wb), dtype=torch.int64, device=self.device)
            dbg.mask = torch.empty((n, Hb, Wb), dtype=torch.uint8, device=self.device)
            dbg.stats = torch.zeros((n, 8), dtype=torch.int32, device=self.device)
            dbg.in_window = torch.zeros(n, dtype=torch.int32, device=self.device)
        ws = self._workspace(n)
        index = self.pano_index(pano_depth)
        P = int(pano_rgb.shape[0])
        ptr = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
        with torch.cuda.device(self.device):
            st = self.lib.salve_bev_render_batch(
                ctypes.byref(self.cfg), ptr(pano_rgb), ptr(pano_depth), P, ptr(self.sphere), ptr(index), ptr(hyps_dev), n,
                ptr(out_bev), ptr(dbg.img_xy), ptr(dbg.keys), ptr(dbg.mask), ptr(dbg.stats), ptr(dbg.in_window), status.ptr(self.device),
                ptr(ws), ws.numel(), self._stream(),
            )
        _lib.check(st, "salve_bev_render_batch")
        return out_bev, dbg

    def scatter(self, pano_rgb: torch.Tensor, pano_depth: torch.Tensor, hyps_dev: torch.Tensor, n: int, out_bev: torch.Tensor,
                in_window: Optional[torch.Tensor] = None) -> None:
        """First half of `render`: the sparse images (z-order winners' colours) into `out_bev`, the occupancy bitmaps into the
        workspace.  `in_window` (int32 [n]) receives the number of points inside the window per render (0 => the reference
        writes no tile, bev_rendering_utils.py:279, 623-627)."""
        ws = self._workspace(n)
        index = self.pano_index(pano_depth)
        with torch.cuda.device(self.device):
            st = self.lib.salve_bev_scatter(
                ctypes.byref(self.cfg), ctypes.c_void_p(pano_rgb.data_ptr()), ctypes.c_void_p(pano_depth.data_ptr()),
                int(pano_rgb.shape[0]), ctypes.c_void_p(self.sphere.data_ptr()), ctypes.c_void_p(index.data_ptr()),
                ctypes.c_void_p(hyps_dev.data_ptr()), n, ctypes.c_void_p(out_bev.data_ptr()), None, None,
                ctypes.c_void_p(0 if in_window is None else in_window.data_ptr()), status.ptr(self.device),
                ctypes.c_void_p(ws.data_ptr()), ws.numel(), self._stream())
        _lib.check(st, "salve_bev_scatter")

    def densify(self, n: int, out_bev: torch.Tensor) -> torch.Tensor:
        """Second half of `render`: completes the sparse images of `scatter` in place (same `out_bev`, same workspace slot)."""
        ws = self._workspace(n)
        with torch.cuda.device(self.device):
            st = self.lib.salve_bev_densify(ctypes.byref(self.cfg), n, ctypes.c_void_p(out_bev.data_ptr()), None, None,
                                            status.ptr(self.device), ctypes.c_void_p(ws.data_ptr()), ws.numel(), self._stream())
        _lib.check(st, "salve_bev_densify")
        return out_bev

    def densify_tiles(self, n: int, out_bev: torch.Tensor, jobs_a: torch.Tensor, jobs_b: torch.Tensor, tiles_b: torch.Tensor, out: torch.Tensor,
                      out_c: int) -> torch.Tensor:
        """`densify` + `tile_pairs(pretiled=True)` as ONE launch (include/salve_hip.h: salve_bev_densify_tiles): every render's workgroup writes
        its verifier tile as soon as the image is complete.  `jobs_a` / `jobs_b` (upload_tile_jobs, the second with pretiled=True) are indexed by
        RENDER of the launch: destination sample and channel of render r / the pair's pretiled second image and its channel."""
        ws = self._workspace(n)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            st = self.lib.salve_bev_densify_tiles(ctypes.byref(self.cfg), n, p(out_bev), p(jobs_a), p(jobs_b), p(tiles_b), p(self.coef_y), p(self.coef_x),
                                                  self.resize, self.crop, p(self.lut), p(out), out_c, status.ptr(self.device), p(ws), ws.numel(), self._stream())
        _lib.check(st, "salve_bev_densify_tiles")
        return out_bev

    def render_counted(self, pano_rgb: torch.Tensor, pano_depth: torch.Tensor, hyps_dev: torch.Tensor, n: int,
                       out_bev: torch.Tensor, counts: torch.Tensor) -> None:
        """`render` that also reports, per render, how many points fell inside the window (int32 [n])."""
        ws = self._workspace(n)
        index = self.pano_index(pano_depth)
        ptr = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
        with torch.cuda.device(self.device):
            st = self.lib.salve_bev_render_batch(
                ctypes.byref(self.cfg), ptr(pano_rgb), ptr(pano_depth), int(pano_rgb.shape[0]), ptr(self.sphere), ptr(index), ptr(hyps_dev), n,
                ptr(out_bev), None, None, None, None, ptr(counts), status.ptr(self.device), ptr(ws), ws.numel(), self._stream())
        _lib.check(st, "salve_bev_render_batch")

    def render_points(self, xyz: np.ndarray, rgb_u8: np.ndarray):
        """One BEV image from an explicit world-frame point cloud (host arrays).  Returns (int32 [1,H,W], n_in_window)."""
        xyz_d = torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.float64)).to(self.device)
        rgb_d = torch.from_numpy(np.ascontiguousarray(rgb_u8, dtype=np.uint8)).to(self.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        Hb, Wb = self.bev_hw
        bev = torch.empty((1, Hb, Wb), dtype=torch.int32, device=self.device)
        ws = self._workspace(1)
        with torch.cuda.device(self.device):
            st = self.lib.salve_bev_scatter_points(ctypes.byref(self.cfg), ctypes.c_void_p(xyz_d.data_ptr()), ctypes.c_void_p(rgb_d.data_ptr()),
                                                   int(xyz_d.shape[0]), ctypes.c_void_p(bev.data_ptr()), ctypes.c_void_p(cnt.data_ptr()),
                                                   ctypes.c_void_p(ws.data_ptr()), ws.numel(), self._stream())
        _lib.check(st, "salve_bev_scatter_points")
        self.densify(1, bev)
        return bev, int(cnt.item())

    def keys_from_pixels(self, xy: torch.Tensor, rgb: torch.Tensor, out_bev: torch.Tensor) -> None:
        """Sparse image of ONE render from explicit integer pixels (int32 [n, 2]) and colours (uint8 [n, 3]) into `out_bev`
        (int32 [1, H, W]); follow with densify(1, out_bev)."""
        ws = self._workspace(1)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            st = self.lib.salve_bev_keys_from_pixels(ctypes.byref(self.cfg), p(xy), p(rgb), int(xy.shape[0]), p(out_bev), p(ws), ws.numel(), self._stream())
        _lib.check(st, "salve_bev_keys_from_pixels")

    def check(self, what: str) -> None:
        """Raise if a kernel of this device reported a failure (a star walk that did not close leaves an incomplete image).
        Synchronises: call where the host is about to read the images anyway."""
        status.check(self.device, what)

    def export_u8(self, bev: torch.Tensor) -> torch.Tensor:
        """int32 [n,H,W] -> uint8 [n,H,W,3] (the array `render_bev_image` returns)."""
        n, Hb, Wb = bev.shape
        out = torch.empty((n, Hb, Wb, 3), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            st = self.lib.salve_bev_export_u8(ctypes.c_void_p(bev.data_ptr()), n, Hb, Wb, ctypes.c_void_p(out.data_ptr()), self._stream())
        _lib.check(st, "salve_bev_export_u8")
        return out

    def upload_tile_jobs(self, bev_index: Sequence[int], slot: Sequence[int], chan: Sequence[int], pretiled: bool = False) -> torch.Tensor:
        """Tile jobs naming image `bev_index` of a BEV array -- or, with pretiled, of an array of TILE_U8X4 images."""
        Hb, Wb = self.bev_hw
        j = np.zeros(len(slot), dtype=_lib.TILE_JOB_DTYPE)
        j["bev_offset"] = np.asarray(bev_index, dtype=np.int64) * ((self.crop * self.crop) if pretiled else (Hb * Wb))
        j["slot"] = np.asarray(slot, dtype=np.int32)
        j["chan"] = np.asarray(chan, dtype=np.int32)
        return torch.from_numpy(j.view(np.uint8)).to(self.device)

    def tiles(self, bev: torch.Tensor, jobs_dev: torch.Tensor, n_jobs: int, out: torch.Tensor, fmt: int, out_c: int) -> torch.Tensor:
        """Resize -> crop -> normalise into `out` (float32 NCHW or fp16 NHWC, see include/salve_hip.h)."""
        Hb, Wb = self.bev_hw
        with torch.cuda.device(self.device):
            st = self.lib.salve_bev_tiles(
                ctypes.c_void_p(bev.data_ptr()), Hb, Wb, ctypes.c_void_p(jobs_dev.data_ptr()), n_jobs,
                ctypes.c_void_p(self.coef_y.data_ptr()), ctypes.c_void_p(self.coef_x.data_ptr()), self.resize, self.crop,
                ctypes.c_void_p(self.lut.data_ptr()), ctypes.c_void_p(out.data_ptr()), fmt, out_c, self._stream(),
            )
        _lib.check(st, "salve_bev_tiles")
        return out

    def pretile(self, bev: torch.Tensor) -> torch.Tensor:
        """int32 [n, H, W] BEV images -> int32 [n, crop, crop] resized + cropped images (TILE_U8X4), for `tile_pairs(pretiled=True)`."""
        n = int(bev.shape[0])
        out = torch.empty((n, self.crop, self.crop), dtype=torch.int32, device=self.device)
        jobs = self.upload_tile_jobs(np.arange(n), np.arange(n), np.zeros(n, dtype=np.int64))
        return self.tiles(bev, jobs, n, out, _lib.TILE_U8X4, 3)

    def tile_pairs(self, bev_a: torch.Tensor, jobs_a: torch.Tensor, bev_b: torch.Tensor, jobs_b: torch.Tensor, n_pairs: int,
                   out: torch.Tensor, out_c: int, pretiled: bool = False) -> torch.Tensor:
        """Both tiles of every early-fusion pair in one pass (fp16 NHWC; include/salve_hip.h: salve_bev_tile_pairs).  pretiled:
        `bev_b` is `pretile`'s output and `jobs_b` was uploaded with pretiled=True."""
        Hb, Wb = self.bev_hw
        with torch.cuda.device(self.device):
            st = self.lib.salve_bev_tile_pairs(
                ctypes.c_void_p(bev_a.data_ptr()), ctypes.c_void_p(bev_b.data_ptr()), Hb, Wb, ctypes.c_void_p(jobs_a.data_ptr()),
                ctypes.c_void_p(jobs_b.data_ptr()), n_pairs, ctypes.c_void_p(self.coef_y.data_ptr()), ctypes.c_void_p(self.coef_x.data_ptr()),
                self.resize, self.crop, ctypes.c_void_p(self.lut.data_ptr()), ctypes.c_void_p(out.data_ptr()), out_c, 1 if pretiled else 0, self._stream(),
            )
        _lib.check(st, "salve_bev_tile_pairs")
        return out

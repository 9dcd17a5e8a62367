"""Fused render -> verify pipeline: alignment hypotheses in, verifier logits out, no JPEG hop.

This is the MI355X counterpart of running the reference's two drivers back to back --
scripts/render_dataset_bev.py:91-117 (one `generate_texture_maps_for_pair` per hypothesis x surface) and
scripts/test.py:155-277 (DataLoader -> model -> softmax) -- with the tiles handed from the rasteriser to the
verifier in HBM instead of through JPEG files (bev_rendering_utils.py:629-630 -> zind_data.py:306-315).

Work decomposition per hypothesis (i1, i2, i2Ti1) and surface:
  * pano i1 is rendered under the pose (hypothesis dependent)          -> one render per hypothesis x surface;
  * pano i2 is rendered at identity (bev_rendering_utils.py:455), which does not depend on the hypothesis, so
    its BEV image is rendered once per (pano, surface) and cached on the device.
Channel order of the early-fusion input follows zind_data.py:306-315: single surface (img1, img2); two surfaces
(ceiling1, ceiling2, floor1, floor2) -- where "1" and "2" are the two tiles of the pair in FILE-NAME order
(zind_data.py:110), which `HypothesisTable.swap` carries (ingest.FloorHypotheses.table computes it from the pano stems).

A render with no point inside the BEV window makes the reference write no tile for that pair
(bev_rendering_utils.py:279-280, 623-627), so the pair never reaches the verifier: `valid_mask` reports those hypotheses
from the in-window counts of the posed render (counted by the scatter kernel) and of the cached identity render.

Multi-GPU: hypotheses are independent, so each rank takes a contiguous block of the table, holds all panoramas and
a full weight replica, and the logits are collected with ONE all-gather (RCCL) -- SURVEY.md section 8e.
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from salve_amd import _lib, status, tracing
from salve_amd.rasteriser import SURFACES, BevRasteriser, pack_hypotheses
from salve_amd.synthetic import HypothesisTable

MODALITY_SURFACES = {
    ("floor_rgb_texture",): ["floor"],
    ("ceiling_rgb_texture",): ["ceiling"],
    ("ceiling_rgb_texture", "floor_rgb_texture"): ["ceiling", "floor"],
}


def surfaces_for(modalities: Sequence[str]) -> List[str]:
    key = tuple(sorted(modalities))
    if key not in MODALITY_SURFACES:
        raise RuntimeError(f"Unsupported modalities for the fused render+verify path: {modalities}")
    return MODALITY_SURFACES[key]


class RenderVerifyPipeline:
    def __init__(self, model, device: torch.device, pano_hw: Tuple[int, int] = (512, 1024), chunk: int = 512,
                 overlap: bool = True, streams: int = 3) -> None:
        self.device = torch.device(device)
        self.model = model
        self.surfaces = surfaces_for(model.modalities)
        self.engine = model.compiled(self.device)
        self.ras = BevRasteriser(self.device, pano_hw=pano_hw)
        self.chunk = chunk
        S = len(self.surfaces)
        Hb, Wb = self.ras.bev_hw
        # overlap=True: two sets of chunk buffers -- the rasteriser (VALU / LDS-latency bound) fills one on its own HIP
        # stream while the verifier (MFMA / HBM bound) consumes the other, so the tail of one launch runs under the other
        # kernel.  (Requires the library to be built without SLP-packed fp32: DESIGN.md section 8.)
        self.nbuf = 2 if overlap else 1
        self.bevs = [torch.empty((chunk * S, Hb, Wb), dtype=torch.int32, device=self.device) for _ in range(self.nbuf)]
        # tiles: fp16 NHWC, pad channels (never written) stay zero
        self.tile_bufs = [torch.zeros((chunk, self.ras.crop, self.ras.crop, self.engine.in_channels), dtype=torch.float16,
                                      device=self.device) for _ in range(self.nbuf)]
        self.bev, self.tiles = self.bevs[0], self.tile_bufs[0]
        self.render_stream = torch.cuda.Stream(self.device) if overlap else None
        # streams = 3: the scatter of chunk i+2 (memory-side atomics, HBM) additionally runs under the densify of chunk
        # i+1 (LDS / VALU) on a stream of its own, with two rasteriser workspaces
        self.scatter_stream = torch.cuda.Stream(self.device) if (overlap and streams >= 3) else None
        self.pano_rgb = self.pano_depth = self.ref_bev = self.ref_in_window = None
        self.n_panos = 0

    # ------------------------------------------------------------------ panoramas
    def load_panos(self, rgb: np.ndarray, depth: np.ndarray) -> None:
        """Upload P panoramas and render their hypothesis-independent (identity) BEV images once."""
        self.set_panos(*self.ras.upload_panos(rgb, depth))

    def set_panos(self, rgb_dev: torch.Tensor, depth_dev: torch.Tensor) -> None:
        """Panoramas already on the device (ingest.PanoStore): uint8 [P,H,W,3], uint16 bits as int16 [P,H,W]."""
        if tuple(rgb_dev.shape[1:3]) != tuple(self.ras.pano_hw) or tuple(depth_dev.shape[1:]) != tuple(self.ras.pano_hw):
            raise RuntimeError(f"panoramas must be {self.ras.pano_hw}, got {tuple(rgb_dev.shape[1:3])} / {tuple(depth_dev.shape[1:])}")
        self.pano_rgb, self.pano_depth = rgb_dev.contiguous(), depth_dev.contiguous()
        P = self.n_panos = int(rgb_dev.shape[0])
        S = len(self.surfaces)
        idx = np.repeat(np.arange(P), S)
        surf = np.tile([SURFACES[s] for s in self.surfaces], P)
        h = pack_hypotheses(idx, surf, np.tile(np.eye(2, dtype=np.float32), (P * S, 1, 1)), np.zeros((P * S, 2), np.float32),
                            np.zeros(P * S))
        Hb, Wb = self.ras.bev_hw
        self.ref_bev = torch.empty((P * S, Hb, Wb), dtype=torch.int32, device=self.device)
        self.ref_in_window = torch.zeros(P * S, dtype=torch.int32, device=self.device)
        hd = self.ras.upload_hypotheses(h)
        with tracing.range("salve.identity_renders"):
            for lo in range(0, P * S, 256):
                n = min(256, P * S - lo)
                self.ras.render_counted(self.pano_rgb, self.pano_depth, hd[lo * _lib.HYP_DTYPE.itemsize:], n, self.ref_bev[lo:lo + n],
                                        self.ref_in_window[lo:lo + n])

    # ------------------------------------------------------------------ hypotheses
    def prepare(self, hyp: HypothesisTable):
        """Upload the render table and the tile job tables of a hypothesis shard (once, outside the timed loop)."""
        N, S = len(hyp), len(self.surfaces)
        surf_ids = [SURFACES[s] for s in self.surfaces]
        rows = pack_hypotheses(np.repeat(hyp.i1, S), np.tile(surf_ids, N), np.repeat(hyp.R, S, axis=0), np.repeat(hyp.t, S, axis=0),
                               np.ones(N * S))
        j = np.arange(N)
        slot = j % self.chunk
        # tile order inside a surface's channel pair: (posed i1, identity i2) unless the file names of the pair sort the
        # other way round (zind_data.py:110)
        swap = np.zeros(N, dtype=np.int64) if hyp.swap is None else np.asarray(hyp.swap).astype(np.int64)
        jobs1_bev, jobs1_slot, jobs1_chan = [], [], []
        jobs2_bev, jobs2_slot, jobs2_chan = [], [], []
        for si in range(S):
            jobs1_bev.append(slot * S + si)           # render output of this chunk
            jobs1_slot.append(slot)
            jobs1_chan.append(6 * si + 3 * swap)
            jobs2_bev.append(hyp.i2.astype(np.int64) * S + si)  # cached identity render of pano i2
            jobs2_slot.append(slot)
            jobs2_chan.append(6 * si + 3 * (1 - swap))
        # job tables are stored hypothesis-major so that a chunk is a contiguous slice
        st = lambda parts: np.stack(parts, 1).reshape(-1)
        return {
            "n": N,
            "i2": np.asarray(hyp.i2).astype(np.int64),
            "rows": self.ras.upload_hypotheses(rows),
            "jobs1": self.ras.upload_tile_jobs(st(jobs1_bev), st(jobs1_slot), st(jobs1_chan)),
            "jobs2": self.ras.upload_tile_jobs(st(jobs2_bev), st(jobs2_slot), st(jobs2_chan)),
            "in_window": torch.zeros(N * S, dtype=torch.int32, device=self.device),  # posed renders, filled by score()
        }

    @staticmethod
    def _timed(timers, units: int, tag: str):
        """(start, end) HIP events on the CURRENT stream, appended to `timers` as (start, end, units, tag); None, None if
        no timing was asked for."""
        if timers is None:
            return None, None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        timers.append((e0, e1, units, tag))
        e0.record()
        return e0, e1

    def _scatter_chunk(self, prepared, lo: int, n: int, slot: int, timers=None) -> None:
        S = len(self.surfaces)
        self.ras.ws_slot = slot
        _, e1 = self._timed(timers, n * S, "scatter")
        with tracing.range("salve.scatter"):
            self.ras.scatter(self.pano_rgb, self.pano_depth, prepared["rows"][lo * S * _lib.HYP_DTYPE.itemsize:], n * S,
                             in_window=prepared["in_window"][lo * S:(lo + n) * S])
        if e1 is not None:
            e1.record()

    def _densify_chunk(self, prepared, lo: int, n: int, buf: int, slot: int, timers=None) -> None:
        S = len(self.surfaces)
        jb = _lib.TILE_JOB_DTYPE.itemsize
        bev, tiles = self.bevs[buf], self.tile_bufs[buf]
        self.ras.ws_slot = slot
        _, e1 = self._timed(timers, n * S, "densify")  # benchmark: HIP events on the stream the kernel is launched on
        with tracing.range("salve.densify"):
            self.ras.densify(n * S, bev)
        if e1 is not None:
            e1.record()
        with tracing.range("salve.tiles"):
            self.ras.tiles(bev, prepared["jobs1"][lo * S * jb:], n * S, tiles, _lib.TILE_F16_NHWC, self.engine.in_channels)
            self.ras.tiles(self.ref_bev, prepared["jobs2"][lo * S * jb:], n * S, tiles, _lib.TILE_F16_NHWC, self.engine.in_channels)

    def _verify_chunk(self, buf: int, n: int, out: torch.Tensor, vtimers=None) -> None:
        e0 = e1 = None
        if vtimers is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        with tracing.range("salve.verify"):
            self.engine.forward_nhwc(self.tile_bufs[buf][:n], out=out)
        if vtimers is not None:
            e1.record()
            vtimers.append((e0, e1, n))

    def score(self, prepared, out: Optional[torch.Tensor] = None, timers=None, vtimers=None) -> torch.Tensor:
        """Render + verify every hypothesis of a prepared shard.  Returns fp32 logits [N, num_classes].
        `timers` / `vtimers` = lists: receive (start event, end event, units, "scatter" | "densify") for the rasteriser
        launches / (start, end, units) for the verifier forward of every chunk (benchmark rooflines; the events are recorded
        on the stream the kernels are launched on)."""
        N = prepared["n"]
        if out is None:
            out = torch.empty((N, self.engine.num_classes), dtype=torch.float32, device=self.device)
        chunks = [(lo, min(self.chunk, N - lo)) for lo in range(0, N, self.chunk)]
        if self.render_stream is None:
            for lo, n in chunks:
                self._scatter_chunk(prepared, lo, n, 0, timers)
                self._densify_chunk(prepared, lo, n, 0, 0, timers)
                self._verify_chunk(0, n, out[lo:lo + n], vtimers)
            return out
        main = torch.cuda.current_stream(self.device)
        self.render_stream.wait_stream(main)
        consumed = [torch.cuda.Event() for _ in chunks]
        densified = [torch.cuda.Event() for _ in chunks]
        if self.scatter_stream is not None:
            self.scatter_stream.wait_stream(main)
            scattered = [torch.cuda.Event() for _ in chunks]
            for i, (lo, n) in enumerate(chunks):
                with torch.cuda.stream(self.scatter_stream):
                    if i >= 2:
                        self.scatter_stream.wait_event(densified[i - 2])  # that densify is done with this workspace
                    self._scatter_chunk(prepared, lo, n, i % 2, timers)
                    scattered[i].record(self.scatter_stream)
                with torch.cuda.stream(self.render_stream):
                    self.render_stream.wait_event(scattered[i])
                    if i >= self.nbuf:
                        self.render_stream.wait_event(consumed[i - self.nbuf])  # the verifier is done with this buffer set
                    self._densify_chunk(prepared, lo, n, i % self.nbuf, i % 2, timers)
                    densified[i].record(self.render_stream)
                main.wait_event(densified[i])
                self._verify_chunk(i % self.nbuf, n, out[lo:lo + n], vtimers)
                consumed[i].record(main)
            self.render_stream.wait_stream(main)
            self.scatter_stream.wait_stream(self.render_stream)
            main.wait_stream(self.scatter_stream)
            self.ras.ws_slot = 0
            return out
        for i, (lo, n) in enumerate(chunks):
            with torch.cuda.stream(self.render_stream):
                if i >= self.nbuf:
                    self.render_stream.wait_event(consumed[i - self.nbuf])  # the verifier is done with this buffer set
                self._scatter_chunk(prepared, lo, n, 0, timers)
                self._densify_chunk(prepared, lo, n, i % self.nbuf, 0, timers)
                densified[i].record(self.render_stream)
            main.wait_event(densified[i])
            self._verify_chunk(i % self.nbuf, n, out[lo:lo + n], vtimers)
            consumed[i].record(main)
        self.render_stream.wait_stream(main)
        return out

    def valid_mask(self, prepared) -> np.ndarray:
        """bool [N] (host): hypotheses for which the reference writes tiles at all -- both renders of every surface have
        at least one point inside the BEV window (bev_rendering_utils.py:279-280, 464-466, 623-627).  Call after score();
        synchronises."""
        S = len(self.surfaces)
        posed = prepared["in_window"].view(-1, S).cpu().numpy() > 0
        ident = self.ref_in_window.view(-1, S).cpu().numpy()[prepared["i2"]] > 0
        return (posed & ident).all(axis=1)

    def check(self, what: str = "render + verify") -> None:
        """Raise if a kernel of this device reported a failure since the last check (device status word; synchronises)."""
        status.check(self.device, what)


def gather_logits(local: torch.Tensor, world: int, total: Optional[int] = None) -> torch.Tensor:
    """The path's only collective: one all-gather of the per-rank fp32 logits (mirrors DataParallel's gather of the model
    outputs, reference train_utils.py:214-215).  Shards of a contiguous block split differ by at most one row
    (HypothesisTable.shard_bounds), so every rank pads its block to ceil(total / world) rows for the single
    `all_gather_into_tensor` and the padding is dropped afterwards.  `total` = rows of the whole table (default: every
    rank holds local.shape[0] rows)."""
    if world == 1:
        return local
    import torch.distributed as dist

    n_local, C = int(local.shape[0]), int(local.shape[1])
    if total is None:
        total = n_local * world
    per = -(-total // world)
    buf = local.contiguous()
    if n_local != per:
        buf = torch.zeros((per, C), dtype=local.dtype, device=local.device)
        buf[:n_local] = local
    out = torch.empty((world * per, C), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    if per * world == total:
        return out
    parts = []
    for r in range(world):
        lo, hi = HypothesisTable.shard_bounds(total, r, world)
        parts.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(parts, 0)

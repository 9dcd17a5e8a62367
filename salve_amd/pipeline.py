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
    # the rasterised-layout modality (early_fusion.py:24-32, 59-60): its two images follow the texture maps' channels
    # (zind_data.py:26: pano 1 ceiling, pano 2 ceiling, pano 1 floor, pano 2 floor, pano 1 layout, pano 2 layout)
    ("layout",): [],
    ("ceiling_rgb_texture", "floor_rgb_texture", "layout"): ["ceiling", "floor"],
}


def surfaces_for(modalities: Sequence[str]) -> List[str]:
    key = tuple(sorted(modalities))
    if key not in MODALITY_SURFACES:
        raise RuntimeError(f"Unsupported modalities for the fused render+verify path: {modalities}")
    return MODALITY_SURFACES[key]


def render_order(i1: np.ndarray, chunk: int) -> Tuple[np.ndarray, np.ndarray]:
    """(order, rank): inside every chunk of `chunk` consecutive hypotheses the renders are issued in the order of their panorama
    (a stable sort by i1) -- order[r] = the hypothesis rendered r-th, rank[j] = the position of hypothesis j.  Chunks keep their
    hypotheses: launches, buffers and logits stay chunk for chunk where they were."""
    i1 = np.asarray(i1).astype(np.int64)
    N = len(i1)
    order = np.concatenate([lo + np.argsort(i1[lo:lo + chunk], kind="stable") for lo in range(0, N, chunk)]) if N else np.zeros(0, np.int64)
    rank = np.empty(N, dtype=np.int64)
    rank[order] = np.arange(N)
    return order, rank


def pick_launch(n_hypotheses: int, per_hypothesis_bytes: int, budget_bytes: int, surfaces: int = 1, granule_renders: int = 512) -> int:
    """Hypotheses per render / verify launch, chosen instead of documented: the LARGEST launch the memory budget allows, because both
    stages run faster per unit in large launches (a launch ends in a tail of partly filled rounds of resident workgroups: densify
    4.7 us per render in launches of 4096 against 8.1 us at 1024; config 5: 18.5 k hypotheses/s at 512 per launch, 20.6 k at 1024,
    22.0 k at 2048 -- DESIGN.md 6).  The whole shard in one launch if it fits; otherwise the shard is cut into the FEWEST launches that
    fit, of equal size (no short last launch), rounded up to whole rounds of the 512 resident densify workgroups (`granule_renders`
    renders = granule / surfaces hypotheses).  The costly-renders-first order of the densify stage pays from 1025 renders per launch
    and costs at exactly two rounds (1024); the library switches it on only above that (bev_render.hip: ORDER_MIN_RENDERS), and this
    chooser never cuts a shard that has more than 1024 renders into launches of 1024 or fewer unless memory forces it.
    Pure host arithmetic (tests/test_host_logic.py)."""
    n = int(n_hypotheses)
    if n <= 0:
        raise ValueError("n_hypotheses must be positive")
    fit = int(budget_bytes) // max(1, int(per_hypothesis_bytes))
    if fit <= 0:
        raise RuntimeError(f"not even one hypothesis ({per_hypothesis_bytes} B) fits the launch budget of {budget_bytes} B")
    if fit >= n:
        return n
    g = max(1, granule_renders // max(1, int(surfaces)))
    launches = -(-n // fit)
    chunk = -(-n // launches)                 # equal launches
    chunk = -(-chunk // g) * g                # whole rounds of resident workgroups
    while chunk > fit and chunk > g:          # rounding up must not leave the budget
        chunk -= g
    return max(1, min(chunk, fit))


class RenderVerifyPipeline:
    LAUNCH_HBM_FRACTION = 0.5   # of the HBM that is free when the pipeline is created: BEV + tile + rasteriser + activation workspaces of one launch

    def __init__(self, model, device: torch.device, pano_hw: Tuple[int, int] = (512, 1024), chunk: Optional[int] = None,
                 overlap: bool = True, streams: int = 3, n_hypotheses: Optional[int] = None, fuse_tiles: bool = True) -> None:
        """chunk: hypotheses per render / verify launch.  None (default): chosen by `pick_launch` from the HBM that is free now --
        the whole shard of `n_hypotheses` rows in one launch if its workspaces fit LAUNCH_HBM_FRACTION of it (a shard of 4096
        hypotheses needs 57 GB with one surface / ResNet-50, 80 GB with two / ResNet-152, of 288), else the fewest equal launches."""
        self.device = torch.device(device)
        # the status word is one per device: a bit an earlier, unchecked caller left behind must be reported as ITS failure,
        # not raised later by this pipeline's check() under the wrong name
        status.check(self.device, "a launch issued before this RenderVerifyPipeline was created")
        self.model = model
        # fuse_tiles (default): the densify kernel writes every render's verifier tile itself (salve_bev_densify_tiles); False: densify, then
        # salve_bev_tile_pairs in a launch of its own (the form of rounds 2-5; same bits: tests/test_gpu_rasteriser.py)
        self.fuse_tiles = bool(fuse_tiles)
        self.surfaces = surfaces_for(model.modalities)
        self.has_layout = "layout" in set(model.modalities)
        self.engine = model.compiled(self.device)
        self.ras = BevRasteriser(self.device, pano_hw=pano_hw)
        S = len(self.surfaces)
        Hb, Wb = self.ras.bev_hw
        nbuf = 2 if overlap else 1
        if chunk is None:
            if n_hypotheses is None:
                raise ValueError("RenderVerifyPipeline(chunk=None) chooses the launch size itself and needs n_hypotheses (rows of the shard)")
            chunk = pick_launch(n_hypotheses, self.per_hypothesis_bytes(nbuf, 2 if (overlap and streams >= 3) else 1),
                                int(torch.cuda.mem_get_info(self.device)[0] * self.LAUNCH_HBM_FRACTION), max(1, S))
        self.chunk = int(chunk)
        # overlap=True: two sets of chunk buffers -- the rasteriser (VALU / LDS-latency bound) fills one on its own HIP
        # stream while the verifier (MFMA / HBM bound) consumes the other, so the tail of one launch runs under the other
        # kernel.  (Requires the library to be built without SLP-packed fp32: DESIGN.md section 8.)
        self.nbuf = nbuf
        self.bevs = [torch.empty((chunk * S, Hb, Wb), dtype=torch.int32, device=self.device) for _ in range(self.nbuf)]
        # tiles: fp16 NHWC, pad channels (never written) stay zero
        self.tile_bufs = [torch.zeros((chunk, self.ras.crop, self.ras.crop, self.engine.in_channels), dtype=torch.float16,
                                      device=self.device) for _ in range(self.nbuf)]
        self.bev, self.tiles = self.bevs[0], self.tile_bufs[0]
        # layout modality: the posed layout images of a chunk (salve_layout_rasterise), one per hypothesis
        self.layout_bevs = [torch.empty((chunk, Hb, Wb), dtype=torch.int32, device=self.device) for _ in range(self.nbuf)] if self.has_layout else None
        self.render_stream = torch.cuda.Stream(self.device) if overlap else None
        # streams = 3: the scatter of chunk i+2 (memory-side atomics, HBM) additionally runs under the densify of chunk
        # i+1 (LDS / VALU) on a stream of its own, with two rasteriser workspaces
        self.scatter_stream = torch.cuda.Stream(self.device) if (overlap and streams >= 3) else None
        self.pano_rgb = self.pano_depth = self.ref_bev = self.ref_tiles = self.ref_in_window = None
        self.n_panos = 0
        # event bookkeeping of score(): running chunk number, last densify per workspace slot, last verifier per buffer set
        self._seq = 0
        self._densified = [None, None]
        self._consumed = [None] * self.nbuf
        self._panos_ready = None
        self.last_chunk_buffer = {}   # chunk index of the last score() call -> buffer set (self.bevs / self.tile_bufs) it used

    def per_hypothesis_bytes(self, nbuf: int = 1, ws_slots: int = 1) -> int:
        """Device bytes one hypothesis of a launch needs: its BEV images and early-fusion tile (per buffer set), the rasteriser's
        workspace per render (per slot; asked of the library: salve_bev_workspace_bytes) and the verifier's activation ping-pong
        buffers per sample (salve_resnet_workspace_bytes)."""
        import ctypes

        S = len(self.surfaces)
        Hb, Wb = self.ras.bev_hw
        lib = self.ras.lib
        ws_render = lib.salve_bev_workspace_bytes(ctypes.byref(self.ras.cfg), 2) - lib.salve_bev_workspace_bytes(ctypes.byref(self.ras.cfg), 1)
        act = lib.salve_resnet_workspace_bytes(self.engine.handle, 2) - lib.salve_resnet_workspace_bytes(self.engine.handle, 1)
        tile = self.ras.crop * self.ras.crop * self.engine.in_channels * 2
        bev = (S + (1 if self.has_layout else 0)) * Hb * Wb * 4
        return int(nbuf * (bev + tile) + ws_slots * S * ws_render + act)

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
        if S == 0:   # layout only: no texture map is rendered, the panoramas are not read
            return
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
        # the identity images' Resize + Crop, once per (panorama, surface): every hypothesis that names the panorama as its second
        # one then only normalises them (salve_bev_tile_pairs, b_pretiled)
        self.ref_tiles = self.ras.pretile(self.ref_bev)
        # render_counted used workspace slot 0 on the current stream: later launches on other streams wait for this event
        self._panos_ready = torch.cuda.Event()
        self._panos_ready.record(torch.cuda.current_stream(self.device))
        self._densified[0] = self._panos_ready

    # ------------------------------------------------------------------ hypotheses
    def prepare(self, hyp: HypothesisTable, layouts=None, order: Optional[np.ndarray] = None):
        """Upload the render table and the tile job tables of a hypothesis shard (once, outside the timed loop).
        `layouts` (salve_amd.layout.FusedLayouts; required iff the model's modalities include "layout"): the posed layout of
        panorama i1 per hypothesis and the own layout of every panorama -- the geometry is posed on the host exactly as
        layout.py / the reference do (bev_rendering_utils.py:82, 90), the pixels are drawn per chunk by salve_layout_rasterise.

        Inside every chunk the renders are issued in the order of their panorama (a stable sort by i1): the workgroups of
        consecutive renders run side by side, and those of one panorama then read its depth blocks, box table and colours
        through the L2s while they are warm (the splat kernel: -10 % at 64 panoramas).  Nothing else changes: the tile jobs
        name each hypothesis's render by its rank, so tiles, logits and in-window counts stay in hypothesis order."""
        N, S = len(hyp), len(self.surfaces)
        surf_ids = [SURFACES[s] for s in self.surfaces]
        i1 = np.asarray(hyp.i1).astype(np.int64)
        if order is None:
            order, rank = render_order(i1, self.chunk)  # rank[j]: position of hypothesis j in the render order of the whole table
        else:   # development (tools/probe/densify_order_probe.py): a caller's render order; it must keep every hypothesis inside its chunk
            order = np.asarray(order).astype(np.int64)
            assert sorted(order.tolist()) == list(range(N)) and bool((order // self.chunk == np.arange(N) // self.chunk).all())
            rank = np.empty(N, dtype=np.int64)
            rank[order] = np.arange(N)
        rows = pack_hypotheses(np.repeat(i1[order], S), np.tile(surf_ids, N), np.repeat(np.asarray(hyp.R)[order], S, axis=0),
                               np.repeat(np.asarray(hyp.t)[order], S, axis=0), np.ones(N * S))
        j = np.arange(N)
        slot = j % self.chunk
        rslot = rank % self.chunk                   # the hypothesis's render inside its chunk's image buffer
        # tile order inside a surface's channel pair: (posed i1, identity i2) unless the file names of the pair sort the
        # other way round (zind_data.py:110)
        swap = np.zeros(N, dtype=np.int64) if hyp.swap is None else np.asarray(hyp.swap).astype(np.int64)
        jobs1_bev, jobs1_slot, jobs1_chan = [], [], []
        jobs2_bev, jobs2_slot, jobs2_chan = [], [], []
        for si in range(S):
            jobs1_bev.append(rslot * S + si)          # render output of this chunk
            jobs1_slot.append(slot)
            jobs1_chan.append(6 * si + 3 * swap)
            jobs2_bev.append(hyp.i2.astype(np.int64) * S + si)  # cached identity render of pano i2
            jobs2_slot.append(slot)
            jobs2_chan.append(6 * si + 3 * (1 - swap))
        # job tables are stored hypothesis-major so that a chunk is a contiguous slice
        st = lambda parts: np.stack(parts, 1).reshape(-1) if parts else np.zeros(0, dtype=np.int64)   # (layout only: no texture jobs)
        # the same jobs indexed by RENDER of a chunk's launch (salve_bev_densify_tiles): render rslot * S + si of its chunk is pair
        # (slot, si); stored chunk after chunk, a chunk's renders contiguous
        rj_slot = np.full(N * S, -1, dtype=np.int64)
        rj_chan_a, rj_chan_b, rj_ident = np.zeros(N * S, dtype=np.int64), np.zeros(N * S, dtype=np.int64), np.zeros(N * S, dtype=np.int64)
        base = (j // self.chunk) * self.chunk * S      # first render of the hypothesis's chunk
        for si in range(S):
            r = base + rslot * S + si
            rj_slot[r] = slot
            rj_chan_a[r] = 6 * si + 3 * swap
            rj_chan_b[r] = 6 * si + 3 * (1 - swap)
            rj_ident[r] = hyp.i2.astype(np.int64) * S + si
        prepared = {
            "n": N,
            "i2": np.asarray(hyp.i2).astype(np.int64),
            "rank": rank,
            "rows": self.ras.upload_hypotheses(rows),
            "jobs1": self.ras.upload_tile_jobs(st(jobs1_bev), st(jobs1_slot), st(jobs1_chan)),
            "jobs2": self.ras.upload_tile_jobs(st(jobs2_bev), st(jobs2_slot), st(jobs2_chan), pretiled=True),
            "rjobs_a": self.ras.upload_tile_jobs(np.zeros(N * S, dtype=np.int64), rj_slot, rj_chan_a),
            "rjobs_b": self.ras.upload_tile_jobs(rj_ident, rj_slot, rj_chan_b, pretiled=True),
            "in_window": torch.zeros(N * S, dtype=torch.int32, device=self.device),  # posed renders IN RENDER ORDER, filled by score()
            "ready": torch.cuda.Event(),   # the tables are on the device: launches on other streams wait for it
        }
        if self.has_layout:
            from salve_amd import layout as layout_mod

            if layouts is None or len(layouts.posed) != N:
                raise RuntimeError('the "layout" modality needs `layouts` (salve_amd.layout.FusedLayouts) with one posed layout per hypothesis')
            if N and int(np.max(hyp.i2)) >= len(layouts.identity):
                raise RuntimeError("a hypothesis names a panorama without an identity layout")
            # pano i2's own layout does not depend on the hypothesis: drawn, resized and cropped once per panorama (like the
            # identity texture maps), then only normalised by the pair kernel
            ident = layout_mod.rasterise_layouts(layouts.identity, self.device)
            prepared["layout_ref_tiles"] = self.ras.pretile(ident)
            prepared["layouts"] = layout_mod.pack_layouts(layouts.posed, self.device)   # hypothesis order: image j of the table
            base = 6 * S
            prepared["jobsL1"] = self.ras.upload_tile_jobs(slot, slot, base + 3 * swap)
            prepared["jobsL2"] = self.ras.upload_tile_jobs(np.asarray(hyp.i2).astype(np.int64), slot, base + 3 * (1 - swap), pretiled=True)
        prepared["ready"].record(torch.cuda.current_stream(self.device))
        return prepared

    def bev_index(self, prepared, j: int) -> Tuple[int, int]:
        """(chunk, first image inside that chunk's BEV buffer) of hypothesis j's posed renders (S consecutive images)."""
        r = int(prepared["rank"][j])
        return r // self.chunk, (r % self.chunk) * len(self.surfaces)

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

    def _scatter_chunk(self, prepared, lo: int, n: int, buf: int, slot: int, timers=None) -> None:
        S = len(self.surfaces)
        if S == 0:
            return
        self.ras.ws_slot = slot
        _, e1 = self._timed(timers, n * S, "scatter")
        with tracing.range("salve.scatter"):
            self.ras.scatter(self.pano_rgb, self.pano_depth, prepared["rows"][lo * S * _lib.HYP_DTYPE.itemsize:], n * S, self.bevs[buf],
                             in_window=prepared["in_window"][lo * S:(lo + n) * S])
        if e1 is not None:
            e1.record()

    def _densify_chunk(self, prepared, lo: int, n: int, buf: int, slot: int, timers=None) -> None:
        S = len(self.surfaces)
        jb = _lib.TILE_JOB_DTYPE.itemsize
        bev, tiles = self.bevs[buf], self.tile_bufs[buf]
        if S > 0:
            self.ras.ws_slot = slot
            _, e1 = self._timed(timers, n * S, "densify")  # benchmark: HIP events on the stream the kernel is launched on
            with tracing.range("salve.densify"):
                if self.fuse_tiles:   # the render's tile leaves the densify kernel itself, while the image is in the L2
                    self.ras.densify_tiles(n * S, bev, prepared["rjobs_a"][lo * S * jb:], prepared["rjobs_b"][lo * S * jb:], self.ref_tiles, tiles,
                                           self.engine.in_channels)
                else:
                    self.ras.densify(n * S, bev)
            if e1 is not None:
                e1.record()
            if not self.fuse_tiles:
                with tracing.range("salve.tiles"):
                    # (jobs1[k] / jobs2[k] are the two halves of one surface's six channels of one sample: prepare())
                    self.ras.tile_pairs(bev, prepared["jobs1"][lo * S * jb:], self.ref_tiles, prepared["jobs2"][lo * S * jb:], n * S, tiles,
                                        self.engine.in_channels, pretiled=True)
        if self.has_layout:
            with tracing.range("salve.layout"):
                lbev = self.layout_bevs[buf]
                prepared["layouts"].rasterise(lo, n, lbev)
                self.ras.tile_pairs(lbev, prepared["jobsL1"][lo * jb:], prepared["layout_ref_tiles"], prepared["jobsL2"][lo * jb:], n, tiles,
                                    self.engine.in_channels, pretiled=True)

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
        """Render + verify every hypothesis of a prepared shard.  Returns fp32 logits [N, num_classes] (complete once the
        CURRENT stream has caught up).  `timers` / `vtimers` = lists: receive (start event, end event, units, "scatter" |
        "densify") for the rasteriser launches / (start, end, units) for the verifier forward of every chunk (benchmark
        rooflines; the events are recorded on the stream the kernels are launched on).

        Streams: the verifier always runs on the current stream; with overlap the scatter and the densify + tile kernels run
        on their own streams, ordered by events only -- a workspace slot is reused once the densify that read it is done, a
        buffer set once the verifier that read it is done -- so the rasteriser of the NEXT chunk, or of the next call, runs
        under the verifier of this one.  Nothing is joined at the end of a call: everything a caller can observe (the
        logits, the in-window counts, the BEV buffers of this call) is complete when the current stream is."""
        N = prepared["n"]
        if out is None:
            out = torch.empty((N, self.engine.num_classes), dtype=torch.float32, device=self.device)
        main = torch.cuda.current_stream(self.device)
        rs = self.render_stream or main
        ss = self.scatter_stream or rs
        n_slots = 2 if self.scatter_stream is not None else 1
        self.last_chunk_buffer = {}
        for ci, lo in enumerate(range(0, N, self.chunk)):
            n = min(self.chunk, N - lo)
            seq = self._seq
            self._seq += 1
            buf, slot = seq % self.nbuf, seq % n_slots
            self.last_chunk_buffer[ci] = buf
            with torch.cuda.stream(ss):
                if ss is not main:
                    ss.wait_event(prepared["ready"])
                    if self._panos_ready is not None:
                        ss.wait_event(self._panos_ready)
                if self._densified[slot] is not None:
                    # the densify + tile kernels that used this workspace slot are done -- and with them the last reader of the
                    # BEV buffer the scatter is about to write its sparse images into (buffer set == slot whenever the
                    # scatter has a stream of its own; otherwise scatter and densify share a stream)
                    ss.wait_event(self._densified[slot])
                self._scatter_chunk(prepared, lo, n, buf, slot, timers)
                scattered = torch.cuda.Event()
                scattered.record(ss)
            with torch.cuda.stream(rs):
                if rs is not ss:
                    rs.wait_event(scattered)
                if self._consumed[buf] is not None:
                    rs.wait_event(self._consumed[buf])        # the verifier that read this buffer set is done
                self._densify_chunk(prepared, lo, n, buf, slot, timers)
                self._densified[slot] = torch.cuda.Event()
                self._densified[slot].record(rs)
            if rs is not main:
                main.wait_event(self._densified[slot])
            self._verify_chunk(buf, n, out[lo:lo + n], vtimers)
            self._consumed[buf] = torch.cuda.Event()
            self._consumed[buf].record(main)
        self.ras.ws_slot = 0
        return out

    def valid_mask(self, prepared) -> np.ndarray:
        """bool [N] (host): hypotheses for which the reference writes tiles at all -- both renders of every surface have
        at least one point inside the BEV window (bev_rendering_utils.py:279-280, 464-466, 623-627).  Call after score();
        synchronises."""
        S = len(self.surfaces)
        if S == 0:   # layout only: rasterize_room_layout_pair always returns both images (bev_rendering_utils.py:48-101)
            return np.ones(prepared["n"], dtype=bool)
        posed = prepared["in_window"].view(-1, S).cpu().numpy()[prepared["rank"]] > 0
        ident = self.ref_in_window.view(-1, S).cpu().numpy()[prepared["i2"]] > 0
        return (posed & ident).all(axis=1)

    def check(self, what: str = "render + verify") -> None:
        """Raise if a kernel of this device reported a failure since the last check (device status word; synchronises)."""
        status.check(self.device, what)


def gather_logits(local: torch.Tensor, world: int, total: Optional[int] = None, force: bool = False,
                  counts: Optional[Sequence[int]] = None) -> torch.Tensor:
    """The path's only collective: one all-gather of the per-rank fp32 logits (mirrors DataParallel's gather of the model
    outputs, reference train_utils.py:214-215).  Shards of a contiguous block split differ by at most one row
    (HypothesisTable.shard_bounds), so every rank pads its block to ceil(total / world) rows for the single
    `all_gather_into_tensor` and the padding is dropped afterwards.  `total` = rows of the whole table (default: every
    rank holds local.shape[0] rows).  `force`: run the collective even in a world of one (bench.py --force-dist: the RCCL
    path on a single GPU).  `counts` (instead of `total`): the rows every rank holds, for blocks that are not the table's block
    split (evaluate.run_test_epoch: blocks of whole batches) -- every rank pads to max(counts)."""
    if world == 1 and not force:
        return local
    import torch.distributed as dist

    n_local, C = int(local.shape[0]), int(local.shape[1])
    if counts is not None:
        assert len(counts) == world and n_local == int(counts[dist.get_rank()]), (counts, n_local)
        per = max(1, max(int(c) for c in counts))
        buf = torch.zeros((per, C), dtype=local.dtype, device=local.device)
        buf[:n_local] = local
        out = torch.empty((world * per, C), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, buf)
        return torch.cat([out[r * per: r * per + int(counts[r])] for r in range(world)], 0)
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

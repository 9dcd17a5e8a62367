"""The step BEFORE the hot path: panoramas and alignment hypotheses from disk to the device.

* `PanoStore` -- the panoramas of one floor, resident on the GPU in the layout the rasteriser reads
  (`u8 [P,512,1024,3]`, `u16 [P,512,1024]`): the RGB JPEG is decoded on the host (Pillow) and resized ON THE DEVICE with
  cv2's INTER_LINEAR arithmetic (bev_rendering_utils.py:370-375; `salve_resize_rgb_u8`); the `.depth.png` is the uint16
  millimetre map HoHoNet inference wrote (infer_depth.py:55-62, read at bev_rendering_utils.py:367).
* `load_floor_hypotheses` -- the work list of one (building, floor): `{root}/{building}/{floor}/{label}/{i1}_{i2}__
  {wdo}_{k1}_{k2}_{configuration}.json` files (producer scripts/export_alignment_hypotheses.py:75-90, 234-238), each a
  Sim(2) `{"R": [4], "t": [2], "s": float}` (sim2.py:180-188), enumerated exactly as scripts/render_dataset_bev.py:80-110
  does: labels in the order gt_alignment_approx, incorrect_alignment; files sorted by path; pair_idx restarts per label.
* `score_floor` -- both together through the fused pipeline, producing the prediction files of evaluate.py.
"""

from __future__ import annotations

import ctypes
import glob
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from salve_amd import _lib
from salve_amd.common.sim2 import Sim2
from salve_amd.rasteriser import linear_resize_taps
from salve_amd.synthetic import HypothesisTable
from salve_amd.utils import image_io
from salve_amd.utils.bev_rendering_utils import bev_fname_from_img_fpath

LABEL_TYPES = ("gt_alignment_approx", "incorrect_alignment")  # scripts/render_dataset_bev.py:85 ; is_match = 1, 0


def panoid_from_fpath(fpath: str) -> int:
    """`floor_01_partial_room_04_pano_5.jpg` -> 5 (scripts/render_dataset_bev.py:27-31)."""
    return int(Path(fpath).stem.split("_")[-1])


def floor_pano_fpaths(raw_dataset_dir: str, building_id: str) -> Dict[int, str]:
    """pano id -> JPEG path for a building (scripts/render_dataset_bev.py:77-78)."""
    return {panoid_from_fpath(p): p for p in glob.glob(f"{raw_dataset_dir}/{building_id}/panos/*.jpg")}


def resize_rgb_on_device(rgb_dev: torch.Tensor, out_hw: Tuple[int, int]) -> torch.Tensor:
    """cv2.resize(..., INTER_LINEAR) of uint8 [n,H,W,3] device images (include/salve_hip.h: salve_resize_rgb_u8)."""
    n, H, W, _ = rgb_dev.shape
    h, w = out_hw
    if (H, W) == (h, w):
        return rgb_dev
    lib = _lib.load()
    out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=rgb_dev.device)
    cy = cx = None
    if not (H == 2 * h and W == 2 * w):
        cy = torch.from_numpy(linear_resize_taps(h, H)).to(rgb_dev.device)
        cx = torch.from_numpy(linear_resize_taps(w, W)).to(rgb_dev.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = lib.salve_resize_rgb_u8(p(rgb_dev.contiguous()), n, H, W, p(out), h, w, p(cy), p(cx),
                                 ctypes.c_void_p(torch.cuda.current_stream(rgb_dev.device).cuda_stream))
    _lib.check(st, "salve_resize_rgb_u8")
    return out


class PanoStore:
    """Panoramas of one floor on the device, addressed by pano id."""

    def __init__(self, device, pano_hw: Tuple[int, int] = (512, 1024)) -> None:
        self.device = torch.device(device)
        self.pano_hw = pano_hw
        self.index: Dict[int, int] = {}
        self.fpaths: List[str] = []
        self.rgb = self.depth = None

    def load(self, img_fpaths: Dict[int, str], depth_save_root: str, building_id: str, pano_ids: Sequence[int]) -> "PanoStore":
        """Decode (host) -> upload -> resize (device).  Depth maps are `{depth_save_root}/{building}/{stem}.depth.png`
        (bev_rendering_utils.py:610-611); one that is missing or of the wrong size raises, as the reference's imread /
        broadcast would."""
        h, w = self.pano_hw
        rgbs, depths = [], []
        for k, pid in enumerate(sorted(set(int(p) for p in pano_ids))):
            fp = img_fpaths[pid]
            rgb = image_io.read_rgb(fp)
            if rgb.ndim == 2:
                rgb = np.repeat(rgb[:, :, None], 3, axis=2)
            depth = image_io.read_depth_png(f"{depth_save_root}/{building_id}/{Path(fp).stem}.depth.png")
            if depth.shape != (h, w):
                raise ValueError(f"depth map must be {w}x{h}, got {depth.shape[::-1]}")
            self.index[pid] = k
            self.fpaths.append(fp)
            rgbs.append(torch.from_numpy(rgb).to(self.device)[None])
            depths.append(depth)
        # panoramas of one tour share a size; resize per distinct size so that mixed inputs still work
        out = [resize_rgb_on_device(t, (h, w)) for t in rgbs]
        self.rgb = torch.cat(out, 0).contiguous()
        self.depth = torch.from_numpy(np.stack(depths).view(np.int16)).to(self.device)
        return self

    def __len__(self) -> int:
        return len(self.index)


@dataclass
class FloorHypotheses:
    """Alignment hypotheses of one floor, in the order the reference renders them."""

    building_id: str
    floor_id: str
    i1: np.ndarray          # [N] pano ids (NOT store indices)
    i2: np.ndarray
    R: np.ndarray           # [N,2,2] float32 (Sim2 stores float32, sim2.py:50-52)
    t: np.ndarray           # [N,2] float32
    s: np.ndarray           # [N] float64 (unused by the renderer, which applies R and t * 1.5 only: :447-451)
    label: np.ndarray       # [N] 1 = gt_alignment_approx, 0 = incorrect_alignment
    pair_idx: np.ndarray    # [N] index within its label directory
    pair_uuid: List[str]    # e.g. door_0_0_identity
    fpaths: List[str]

    def __len__(self) -> int:
        return len(self.fpaths)

    def swap(self, img_fpaths: Dict[int, str]) -> np.ndarray:
        """bool [N]: True where the tile of pano i2 sorts BEFORE the tile of pano i1.  The reference's dataset orders the
        two tiles of a pair by file name (salve/dataset/zind_data.py:110 `pair_fpaths.sort()`); the names differ only in
        the pano stem (bev_rendering_utils.py:582-595), e.g. `..partial_room_07_pano_5` > `..partial_room_04_pano_8` and
        `pano_10` < `pano_9` -- not the order of (i1, i2).  The verifier's first image is the one whose name sorts first."""
        return np.array([Path(img_fpaths[int(a)]).stem > Path(img_fpaths[int(b)]).stem for a, b in zip(self.i1, self.i2)], dtype=bool)

    def table(self, store: PanoStore, img_fpaths: Optional[Dict[int, str]] = None) -> HypothesisTable:
        """Rows for pipeline.RenderVerifyPipeline.prepare: pano ids -> store indices, tile order from the file names."""
        ix = lambda ids: np.array([store.index[int(p)] for p in ids], dtype=np.int32)
        if img_fpaths is None:
            img_fpaths = {pid: store.fpaths[k] for pid, k in store.index.items()}
        return HypothesisTable(i1=ix(self.i1), i2=ix(self.i2), R=self.R.astype(np.float32), t=self.t.astype(np.float32),
                               theta_deg=np.degrees(np.arctan2(self.R[:, 1, 0], self.R[:, 0, 0])).astype(np.float64),
                               swap=self.swap(img_fpaths))

    def tile_paths(self, bev_save_root: str, img_fpaths: Dict[int, str], surface_type: str = "floor") -> List[Tuple[str, str]]:
        """(path of pano i1's tile, path of pano i2's tile) per hypothesis: where generate_texture_maps_for_pair writes the
        posed and the identity render (bev_rendering_utils.py:579-595, 629-630)."""
        out = []
        for j in range(len(self)):
            d = f"{bev_save_root}/{LABEL_TYPES[0] if self.label[j] else LABEL_TYPES[1]}/{self.building_id}"
            out.append(tuple(f"{d}/{bev_fname_from_img_fpath(int(self.pair_idx[j]), self.pair_uuid[j], surface_type, img_fpaths[int(p)])}"
                             for p in (self.i1[j], self.i2[j])))
        return out

    def tile_names(self, bev_save_root: str, img_fpaths: Dict[int, str], surface_type: str = "floor") -> List[Tuple[str, str]]:
        """(fp0, fp1) per hypothesis: the same two paths in the SORTED order in which the reference's dataset hands the tiles
        to the verifier and scripts/test.py writes them to the prediction files (zind_data.py:110, 306-315)."""
        return [tuple(sorted(pair)) for pair in self.tile_paths(bev_save_root, img_fpaths, surface_type)]


def load_floor_hypotheses(hypotheses_save_root: str, building_id: str, floor_id: str) -> FloorHypotheses:
    i1, i2, R, t, s, label, pair_idx, uuid, fps = [], [], [], [], [], [], [], [], []
    for label_type in LABEL_TYPES:
        pairs = sorted(glob.glob(f"{hypotheses_save_root}/{building_id}/{floor_id}/{label_type}/*.json"))
        for k, fp in enumerate(pairs):
            stem = Path(fp).stem
            a, b = stem.split("_")[:2]                      # bev_rendering_utils.py:570-571
            S = Sim2.from_json(fp)
            i1.append(int(a)); i2.append(int(b))
            R.append(np.asarray(S.rotation, dtype=np.float32)); t.append(np.asarray(S.translation, dtype=np.float32)); s.append(float(S.scale))
            label.append(1 if label_type == LABEL_TYPES[0] else 0)
            pair_idx.append(k)
            uuid.append(stem.split("__")[-1])               # :577
            fps.append(fp)
    n = len(fps)
    return FloorHypotheses(building_id, floor_id, np.array(i1, dtype=np.int64), np.array(i2, dtype=np.int64),
                           np.array(R, dtype=np.float32).reshape(n, 2, 2), np.array(t, dtype=np.float32).reshape(n, 2),
                           np.array(s, dtype=np.float64), np.array(label, dtype=np.int64), np.array(pair_idx, dtype=np.int64), uuid, fps)


def score_floor(model, device, raw_dataset_dir: str, depth_save_root: str, hypotheses_save_root: str, bev_save_root: str,
                building_id: str, floor_id: str, serialization_save_dir: str, batch_size: int = 64, chunk: Optional[int] = None):
    """Disk -> predictions for one floor without writing a tile: the fused counterpart of running
    scripts/render_dataset_bev.py and then scripts/test.py on that floor.  chunk = hypotheses per launch; None: the pipeline picks the
    largest launch that fits the free HBM (pipeline.pick_launch)."""
    from salve_amd import evaluate
    from salve_amd.pipeline import RenderVerifyPipeline

    hyps = load_floor_hypotheses(hypotheses_save_root, building_id, floor_id)
    if len(hyps) == 0:
        return None
    img_fpaths = floor_pano_fpaths(raw_dataset_dir, building_id)
    store = PanoStore(device).load(img_fpaths, depth_save_root, building_id, np.concatenate([hyps.i1, hyps.i2]))
    pipe = RenderVerifyPipeline(model, device, pano_hw=store.pano_hw, chunk=chunk, n_hypotheses=len(hyps))
    pipe.set_panos(store.rgb, store.depth)
    return evaluate.run_fused_epoch(pipe, hyps.table(store, img_fpaths), hyps.tile_names(bev_save_root, img_fpaths), hyps.label,
                                    serialization_save_dir, batch_size=batch_size)

"""Batched counterpart of scripts/render_dataset_bev.py: all texture maps of a floor / a building / a split, written as
the JPEG tiles the dataset reader (dataset/zind_data.py) expects.

Where the reference maps `generate_texture_maps_for_pair` over a multiprocessing.Pool (one pair, one surface, two renders
and two file writes per task, scripts/render_dataset_bev.py:86-117), this driver loads the floor's panoramas onto the GPU
once (ingest.PanoStore), renders every (hypothesis, surface) of the floor in large launches -- the identity render of a
panorama once per surface instead of once per hypothesis -- and only then goes back to the host to encode JPEGs.  Same
work list order, same file names, same skip-if-both-exist restart rule, same "nothing is written if either cloud has no
point inside the window" rule (bev_rendering_utils.py:621-627).
"""

from __future__ import annotations

import glob
import os
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from salve_amd import ingest
from salve_amd.dataset.zind_partition import DATASET_SPLITS
from salve_amd.rasteriser import SURFACES, BevRasteriser, pack_hypotheses
from salve_amd.utils import image_io

SURFACE_TYPES = ("floor", "ceiling")  # scripts/render_dataset_bev.py:91


def available_floors(hypotheses_save_root: str, building_id: str) -> List[str]:
    """Floor ids of a building.  The reference asks the ZInD pose graph (posegraph2d.compute_available_floors_for_building,
    scripts/render_dataset_bev.py:166-168); here the floors are those that have alignment hypotheses on disk, which
    is the set the rendering loop does anything for."""
    return sorted(Path(p).name for p in glob.glob(f"{hypotheses_save_root}/{building_id}/floor_*") if Path(p).is_dir())


def render_building_floor_pairs(depth_save_root: str, bev_save_root: str, hypotheses_save_root: str, raw_dataset_dir: str,
                                building_id: str, floor_id: str, layout_save_root: Optional[str], render_modalities: List[str],
                                multiprocess_building_panos: bool = False, num_processes: int = 1, device=None, batch: int = 256,
                                floor_pose_graph=None) -> int:
    """All floor + ceiling texture maps of one floor (scripts/render_dataset_bev.py:34-117).  `multiprocess_building_panos`
    and `num_processes` are accepted for signature compatibility; the parallelism is the GPU's.  Returns the number of
    JPEG files written.
    The "layout" modality needs the floor's pose graph with room layouts and W/D/O objects (the reference loads it with
    hnet_prediction_loader, :61-75, which is outside this path): pass it as `floor_pose_graph`."""
    if "layout" in render_modalities and floor_pose_graph is None:
        raise NotImplementedError("the layout modality needs `floor_pose_graph` (loading inferred layouts is outside this path)")
    hyps = ingest.load_floor_hypotheses(hypotheses_save_root, building_id, floor_id)
    if len(hyps) == 0:
        return 0
    img_fpaths = ingest.floor_pano_fpaths(raw_dataset_dir, building_id)
    written = 0
    if "layout" in render_modalities:
        written += _render_floor_layouts(hyps, img_fpaths, layout_save_root, floor_pose_graph, device)
    if "rgb_texture" not in render_modalities:
        return written
    names = {s: hyps.tile_paths(bev_save_root, img_fpaths, s) for s in SURFACE_TYPES}  # (tile of i1, tile of i2)
    todo = [(j, s) for j in range(len(hyps)) for s in SURFACE_TYPES
            if not (Path(names[s][j][0]).exists() and Path(names[s][j][1]).exists())]   # both exist: skip (idempotent restart)
    if not todo:
        return written
    dev = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
    needed = sorted({int(p) for j, _ in todo for p in (hyps.i1[j], hyps.i2[j])})
    store = ingest.PanoStore(dev).load(img_fpaths, depth_save_root, building_id, needed)
    ras = BevRasteriser(dev, pano_hw=store.pano_hw)
    Hb, Wb = ras.bev_hw

    def render(rows: List[Tuple[int, str, np.ndarray, np.ndarray, int]]):
        """rows of (store index, surface, R, t, apply_pose) -> (uint8 [n,H,W,3] on the host, in-window point counts)."""
        out, cnts = [], []
        for lo in range(0, len(rows), batch):
            part = rows[lo:lo + batch]
            h = pack_hypotheses([r[0] for r in part], [SURFACES[r[1]] for r in part], np.stack([r[2] for r in part]),
                                np.stack([r[3] for r in part]), [r[4] for r in part])
            counts = torch.zeros(len(part), dtype=torch.int32, device=dev)
            bev = torch.empty((len(part), Hb, Wb), dtype=torch.int32, device=dev)
            ras.render_counted(store.rgb, store.depth, ras.upload_hypotheses(h), len(part), bev, counts)
            out.append(ras.export_u8(bev).cpu().numpy())
            cnts.append(counts.cpu().numpy())
            ras.check(f"render_building_floor_pairs({building_id}, {floor_id})")   # before any of these images is written to disk
        return np.concatenate(out), np.concatenate(cnts)

    eye, zero = np.eye(2, dtype=np.float32), np.zeros(2, dtype=np.float32)
    # pano i2 is rendered at identity (bev_rendering_utils.py:455): once per (pano, surface)
    ident_keys = sorted({(int(hyps.i2[j]), s) for j, s in todo})
    ident_img, ident_cnt = render([(store.index[p], s, eye, zero, 0) for p, s in ident_keys])
    ident = {k: (ident_img[i], int(ident_cnt[i])) for i, k in enumerate(ident_keys)}
    posed_img, posed_cnt = render([(store.index[int(hyps.i1[j])], s, hyps.R[j], hyps.t[j], 1) for j, s in todo])
    for k, (j, s) in enumerate(todo):
        img2, cnt2 = ident[(int(hyps.i2[j]), s)]
        if int(posed_cnt[k]) == 0 or cnt2 == 0:
            continue  # render_bev_pair -> (None, None): nothing is written for this pair and surface
        fp1, fp2 = names[s][j]
        os.makedirs(os.path.dirname(fp1), exist_ok=True)
        image_io.write_jpeg(fp1, posed_img[k])
        image_io.write_jpeg(fp2, img2)
        written += 2
    return written


def _render_floor_layouts(hyps, img_fpaths: Dict[int, str], layout_save_root: str, floor_pose_graph, device) -> int:
    """Layout tiles of every hypothesis of a floor in one launch (bev_rendering_utils.py:632-663: `floor` names only, skip if
    both files exist).  The layout of pano i2 does not depend on the hypothesis: rendered once per panorama."""
    from salve_amd import layout
    from salve_amd.common.sim2 import Sim2
    from salve_amd.rasteriser import BevRasteriser

    paths = hyps.tile_paths(layout_save_root, img_fpaths, "floor")
    todo = [j for j in range(len(hyps)) if not (Path(paths[j][0]).exists() and Path(paths[j][1]).exists())]
    if not todo:
        return 0
    dev = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
    specs, ident_of = [], {}
    for j in todo:
        S = Sim2(hyps.R[j], hyps.t[j], float(hyps.s[j]))
        s1, s2 = layout.layout_pair_specs(S, floor_pose_graph, int(hyps.i1[j]), int(hyps.i2[j]))
        specs.append(s1)
        ident_of.setdefault(int(hyps.i2[j]), s2)
    ident_ids = sorted(ident_of)
    imgs = layout.rasterise_layouts(specs + [ident_of[p] for p in ident_ids], dev)
    u8 = BevRasteriser(dev).export_u8(imgs).cpu().numpy()
    written = 0
    for k, j in enumerate(todo):
        fp1, fp2 = paths[j]
        os.makedirs(os.path.dirname(fp1), exist_ok=True)
        image_io.write_jpeg(fp1, u8[k])
        image_io.write_jpeg(fp2, u8[len(todo) + ident_ids.index(int(hyps.i2[j]))])
        written += 2
    return written


def render_pairs(num_processes: int, depth_save_root: str, bev_save_root: str, raw_dataset_dir: str, hypotheses_save_root: str,
                 layout_save_root: Optional[str], render_modalities: List[str], split: Optional[str], building_id: Optional[str],
                 multiprocess_building_panos: bool = False, device=None, rank: int = 0, world: int = 1) -> int:
    """All floors of a split's buildings, or of one building (scripts/render_dataset_bev.py:120-191): exactly one of
    `split` / `building_id`; building 1348 is skipped (two panoramas share an id, :160-162).
    world > 1: one process per GPU.  The reference hands its (building, floor) work list to a multiprocessing.Pool
    (`p.starmap(render_building_floor_pairs, args)`, :186-188); here rank r takes items r, r + world, ... of the same list (round
    robin: buildings differ a lot in size, neighbours in the sorted list less so) and writes their files -- the items are
    independent, nothing is exchanged.  Returns the number of JPEG files THIS rank wrote."""
    if building_id is not None and split is not None:
        raise ValueError("Either `split` or `building_id` should be provided, but not both.")
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside a world of {world}")
    written = 0
    for bid, floor_id in floor_work_list(hypotheses_save_root, split, building_id)[rank::world]:
        written += render_building_floor_pairs(depth_save_root, bev_save_root, hypotheses_save_root, raw_dataset_dir, bid, floor_id,
                                               layout_save_root, render_modalities, multiprocess_building_panos, num_processes, device)
    return written


def floor_work_list(hypotheses_save_root: str, split: Optional[str], building_id: Optional[str]) -> List[Tuple[str, str]]:
    """The (building, floor) items of scripts/render_dataset_bev.py:151-184 in its order: buildings sorted, 1348 left out,
    floors in the order `available_floors` lists them."""
    building_ids = sorted(DATASET_SPLITS[split]) if split is not None else [building_id]
    return [(bid, floor_id) for bid in building_ids if bid != "1348" for floor_id in available_floors(hypotheses_save_root, bid)]

"""Bird's-eye-view texture-map rendering with the reference's API, executed by the HIP rasteriser.

Mirror of salve/utils/bev_rendering_utils.py (the texture-map half: :38-45, :254-328, :347-630).  Signatures,
return conventions (None / (None, None)), exceptions and output file names follow the reference so that
scripts/render_dataset_bev.py can call these functions unchanged.  The rasterised-LAYOUT modality (:48-251) runs on the
GPU too (salve_amd/layout.py, csrc/layout.hip); OpenCV's own anti-aliasing arithmetic is unpinned (oracle/layout_oracle.py).

What runs where: file decoding (PIL) and the 2x pano down-scale are host-side ingest; back-projection, pose,
pixel indices, z-order, densification, mask and flip run on the GPU (salve_amd/csrc/bev_render.hip).  There is
no CPU implementation of the rendering here: without the HIP library these functions raise.
"""

from __future__ import annotations

import os
from argparse import Namespace
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

from salve_amd.common.bevparams import BEVParams
from salve_amd.common.sim2 import Sim2
from salve_amd.utils import image_io
from salve_amd.utils.hohonet_pano_utils import get_uni_sphere_xyz

HOHO_S_ZIND_SCALE_FACTOR = 1.5
PANO_W, PANO_H = 1024, 512  # the reference hard-codes the working resolution (:373-374)

_RASTERISERS: Dict[tuple, object] = {}


def _device():
    import torch

    if not torch.cuda.is_available():
        from salve_amd._lib import SalveHipError

        raise SalveHipError("BEV rendering needs the MI355X (HIP device); salve_amd has no CPU renderer")
    return torch.device("cuda", torch.cuda.current_device())


def _rasteriser(bev_params: BEVParams, crop_ratio: float = 80 / 512, scale: float = 0.001):
    from salve_amd.rasteriser import BevRasteriser

    dev = _device()
    key = (str(dev), bev_params.img_h, bev_params.img_w, bev_params.meters_per_px, float(crop_ratio), float(scale))
    if key not in _RASTERISERS:
        _RASTERISERS[key] = BevRasteriser(dev, pano_hw=(PANO_H, PANO_W), bev_params=bev_params, crop_ratio=crop_ratio,
                                          depth_scale=scale)
    return _RASTERISERS[key]


def prune_to_2d_bbox(pts: np.ndarray, rgb: np.ndarray, xmin: float, ymin: float, xmax: float, ymax: float):
    """Keep the points with xmin <= x <= xmax and ymin <= y <= ymax (boundaries included)."""
    x, y = pts[:, 0], pts[:, 1]
    keep = (xmin <= x) & (x <= xmax) & (ymin <= y) & (y <= ymax)
    return pts[keep], rgb[keep]


def render_bev_image(bev_params: BEVParams, xyzrgb: np.ndarray, is_semantics: bool) -> Optional[np.ndarray]:
    """Coloured point cloud [N,6] (world frame, rgb in [0,1]) -> uint8 [H+1, W+1, 3] dense texture map, or None
    when no point falls inside the window (reference :254-328)."""
    if is_semantics:
        raise NotImplementedError("semantic (nearest-neighbour) rendering is not part of the accelerated path")
    ras = _rasteriser(bev_params)
    rgb_u8 = (xyzrgb[:, 3:] * 255).astype(np.uint8)  # the reference's float -> uint8 store truncates (:307-308)
    bev, n_in_window = ras.render_points(xyzrgb[:, :3], rgb_u8)
    if n_in_window == 0:
        return None
    out = ras.export_u8(bev)[0].cpu().numpy()
    ras.check("render_bev_image")   # device status word: a star walk that did not close must not produce a silent, incomplete image
    return out


def grayscale_to_color(gray_img: np.ndarray) -> np.ndarray:
    return np.repeat(gray_img[:, :, None], 3, axis=2).astype(np.uint8)


def _check_args(args) -> None:
    if "crop_ratio" not in args.__dict__:
        raise ValueError("Crop ratio for panorama top and bottom must be provided as `args.crop_ratio`.")
    if "crop_z_range" not in args.__dict__:
        raise ValueError("Z-coordinate range for cropping must be provided as `args.crop_z_range`.")


def _load_pano(depth_fpath: str, rgb_fpath: str) -> Tuple[np.ndarray, np.ndarray]:
    depth = image_io.read_depth_png(depth_fpath)
    rgb = image_io.read_rgb(rgb_fpath)
    if rgb.ndim == 2:
        rgb = grayscale_to_color(rgb)
    rgb = image_io.resize_linear_u8(rgb, (PANO_H, PANO_W))
    if depth.shape != (PANO_H, PANO_W):
        raise ValueError(f"depth map must be {PANO_W}x{PANO_H}, got {depth.shape[::-1]}")
    return rgb, depth


def get_xyzrgb_from_depth(args: Union[SimpleNamespace, Namespace], depth_fpath: str, rgb_fpath: str,
                          is_semantics: bool) -> np.ndarray:
    """Back-projected coloured point cloud [M,6] (reference :347-414).  Host-side helper kept for the reference's
    visualisation scripts; the renderers below do NOT go through it (the GPU back-projects panoramas itself)."""
    _check_args(args)
    if is_semantics:
        raise NotImplementedError("semantic panoramas are not part of the accelerated path")
    rgb, depth_u16 = _load_pano(depth_fpath, rgb_fpath)
    depth = depth_u16[..., None].astype(np.float32) * np.float32(args.scale)
    xyz = depth * get_uni_sphere_xyz(PANO_H, PANO_W)
    xyzrgb = np.concatenate([xyz, rgb / 255.0], 2)
    if args.crop_ratio > 0:
        assert args.crop_ratio < 1
        crop = int(PANO_H * args.crop_ratio)
        xyzrgb = xyzrgb[crop:-crop]
    xyzrgb = xyzrgb.reshape(-1, 6)
    keep = (xyzrgb[:, 2] > args.crop_z_range[0]) & (xyzrgb[:, 2] <= args.crop_z_range[1])
    return xyzrgb[keep]


def _surface_of(crop_z_range) -> int:
    lo, hi = float(crop_z_range[0]), float(crop_z_range[1])
    if lo == -float("inf") and hi == -1.0:
        return 0
    if lo == 0.5 and hi == float("inf"):
        return 1
    raise ValueError(f"crop_z_range {crop_z_range} is neither the floor (-inf, -1.0] nor the ceiling (0.5, inf) range")


def render_bev_pair(args, building_id: str, floor_id: str, i1: int, i2: int, i2Ti1: Sim2, is_semantics: bool):
    """Render pano i1 (moved into i2's frame by i2Ti1) and pano i2 (reference :417-480).  Returns (img1, img2),
    or (None, None) if either cloud has no point inside the window."""
    import torch

    from salve_amd.rasteriser import pack_hypotheses

    _check_args(args)
    if is_semantics:
        raise NotImplementedError("semantic panoramas are not part of the accelerated path")
    ras = _rasteriser(BEVParams(), crop_ratio=args.crop_ratio, scale=args.scale)
    rgb1, d1 = _load_pano(args.depth_i1, args.img_i1)
    rgb2, d2 = _load_pano(args.depth_i2, args.img_i2)
    surface = _surface_of(args.crop_z_range)
    d_rgb, d_depth = ras.upload_panos(np.stack([rgb1, rgb2]), np.stack([d1, d2]))
    R = np.stack([i2Ti1.rotation, np.eye(2, dtype=np.float32)])
    t = np.stack([i2Ti1.translation, np.zeros(2, dtype=np.float32)])
    hyps = ras.upload_hypotheses(pack_hypotheses([0, 1], [surface, surface], R, t, [1, 0]))
    counts = torch.zeros(2, dtype=torch.int32, device=ras.device)
    bev = torch.empty((2,) + ras.bev_hw, dtype=torch.int32, device=ras.device)
    ras.render_counted(d_rgb, d_depth, hyps, 2, bev, counts)
    if int(counts.min().item()) == 0:
        return None, None
    out = ras.export_u8(bev).cpu().numpy()
    ras.check("render_bev_pair")
    return out[0], out[1]


def get_bev_pair_xyzrgb(args, building_id: str, floor_id: str, i1: int, i2: int, i2Ti1: Sim2, is_semantics: bool):
    """The two posed point clouds (reference :483-522); host-side helper for visualisation."""
    from salve_amd.utils.rotation_utils import rotmat2d

    xyzrgb1 = get_xyzrgb_from_depth(args, depth_fpath=args.depth_i1, rgb_fpath=args.img_i1, is_semantics=is_semantics)
    xyzrgb2 = get_xyzrgb_from_depth(args, depth_fpath=args.depth_i2, rgb_fpath=args.img_i2, is_semantics=is_semantics)
    R = rotmat2d(-90)
    xyzrgb1[:, :2] = xyzrgb1[:, :2] @ R.T
    xyzrgb2[:, :2] = xyzrgb2[:, :2] @ R.T
    xyzrgb1[:, :2] = (xyzrgb1[:, :2] @ i2Ti1.rotation.T) + (i2Ti1.translation * HOHO_S_ZIND_SCALE_FACTOR)
    return xyzrgb1, xyzrgb2


def bev_fname_from_img_fpath(pair_idx: int, pair_uuid: str, surface_type: str, img_fpath: str) -> str:
    """`pair_{idx}___{uuid}_{surface}_rgb_{pano stem}.jpg` (reference :582-589); parsed back by
    salve/dataset/zind_data.py:53-58 and salve/common/edge_classification.py:145-175."""
    return f"pair_{pair_idx}___{pair_uuid}_{surface_type}_rgb_{Path(img_fpath).stem}.jpg"


def generate_texture_maps_for_pair(
    img_fpaths_dict: Dict[int, str],
    surface_type: str,
    pair_fpath: str,
    pair_idx: int,
    label_type: str,
    bev_save_root,
    building_id: str,
    floor_id: str,
    depth_save_root: str,
    render_modalities: List[str],
    layout_save_root: str,
    floor_pose_graph,
) -> None:
    """Render and save the two texture maps of one alignment hypothesis and surface (reference :525-630).
    Same 12 positional arguments (picklable for Pool.starmap), same file names, same skip-if-exists restart rule."""
    if surface_type == "floor":
        crop_z_range = [-float("inf"), -1.0]
    elif surface_type == "ceiling":
        crop_z_range = [0.5, float("inf")]
    else:
        raise ValueError(f"unknown surface type {surface_type}")
    i2Ti1 = Sim2.from_json(json_fpath=pair_fpath)
    i1, i2 = (int(v) for v in Path(pair_fpath).stem.split("_")[:2])
    img1_fpath, img2_fpath = img_fpaths_dict[i1], img_fpaths_dict[i2]
    pair_uuid = Path(pair_fpath).stem.split("__")[-1]
    save_dir = f"{bev_save_root}/{label_type}/{building_id}"
    os.makedirs(save_dir, exist_ok=True)
    bev_fpath1 = f"{save_dir}/{bev_fname_from_img_fpath(pair_idx, pair_uuid, surface_type, img1_fpath)}"
    bev_fpath2 = f"{save_dir}/{bev_fname_from_img_fpath(pair_idx, pair_uuid, surface_type, img2_fpath)}"

    if "rgb_texture" in render_modalities:
        depth1 = f"{depth_save_root}/{building_id}/{Path(img1_fpath).stem}.depth.png"
        depth2 = f"{depth_save_root}/{building_id}/{Path(img2_fpath).stem}.depth.png"
        for d in (depth1, depth2):
            if not Path(d).exists():
                # the reference would run HoHoNet here (hohonet_inference.infer_depth_if_nonexistent); monocular depth
                # inference is upstream of this path
                raise FileNotFoundError(f"depth map {d} not found: run the depth-inference stage first")
        args = SimpleNamespace(img_i1=img1_fpath, img_i2=img2_fpath, depth_i1=depth1, depth_i2=depth2, scale=0.001,
                               crop_ratio=80 / 512, crop_z_range=crop_z_range)
        if Path(bev_fpath1).exists() and Path(bev_fpath2).exists():
            return  # both images already exist: idempotent restart
        bev_img1, bev_img2 = render_bev_pair(args, building_id, floor_id, i1, i2, i2Ti1, is_semantics=False)
        if bev_img1 is None or bev_img2 is None:
            return
        image_io.write_jpeg(bev_fpath1, bev_img1)
        image_io.write_jpeg(bev_fpath2, bev_img2)

    if "layout" not in render_modalities:
        return
    # Rasterised layout (:632-663): only for `floor` (the ceiling rendering would be identical), same file names under
    # layout_save_root, same skip-if-both-exist rule.
    if surface_type != "floor":
        return
    building_layout_save_dir = f"{layout_save_root}/{label_type}/{building_id}"
    os.makedirs(building_layout_save_dir, exist_ok=True)
    layout_fpath1 = f"{building_layout_save_dir}/{Path(bev_fpath1).name}"
    layout_fpath2 = f"{building_layout_save_dir}/{Path(bev_fpath2).name}"
    if Path(layout_fpath1).exists() and Path(layout_fpath2).exists():
        return
    layoutimg1, layoutimg2 = rasterize_room_layout_pair(i2Ti1, floor_pose_graph, building_id, floor_id, i1, i2)
    image_io.write_jpeg(layout_fpath1, layoutimg1)
    image_io.write_jpeg(layout_fpath2, layoutimg2)


def rasterize_single_layout(bev_params: BEVParams, room_vertices: np.ndarray, wdo_objs, render_mask: bool = True) -> np.ndarray:
    """Room layout in white, windows / doors / openings as thick coloured segments, flipped vertically (reference :104-156).
    `wdo_objs`: objects with `.type` and `.vertices_local_2d`."""
    from salve_amd import layout

    spec = (np.asarray(room_vertices, dtype=np.float64), [(w.type, np.asarray(w.vertices_local_2d, dtype=np.float64)) for w in wdo_objs])
    img = layout.rasterise_layouts([spec], _device(), bev_params, render_mask=render_mask)
    return _rasteriser(bev_params).export_u8(img).cpu().numpy()[0]


def rasterize_room_layout_pair(i2Ti1: Sim2, floor_pose_graph, building_id: str, floor_id: str, i1: int, i2: int) -> Tuple[np.ndarray, np.ndarray]:
    """BEV rasterisation of the layouts of panoramas i1 (moved into i2's frame by i2Ti1) and i2 (reference :48-101)."""
    from salve_amd import layout

    bp = BEVParams()
    s1, s2 = layout.layout_pair_specs(i2Ti1, floor_pose_graph, i1, i2)
    img = layout.rasterise_layouts([s1, s2], _device(), bp)
    out = _rasteriser(bp).export_u8(img).cpu().numpy()
    return out[0], out[1]

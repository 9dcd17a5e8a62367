"""ctypes binding of libsalve_hip.so (C ABI: include/salve_hip.h).

The product path has NO fallback: if the HIP library has not been built
(`python -c "import __graft_entry__ as g; g.build()"`) every entry point raises.
"""

from __future__ import annotations

import ctypes
from pathlib import Path

import numpy as np

import os

# SALVE_HIP_LIB: development override (ablation builds of tools/probe/build_ablations.sh); the product loads the in-tree library
LIB_PATH = Path(os.environ.get("SALVE_HIP_LIB") or (Path(__file__).resolve().parent / "libsalve_hip.so"))

SALVE_OK = 0
EXPECTED_ABI = 6          # include/salve_hip.h: SALVE_HIP_ABI_VERSION
TILE_F32_NCHW = 0
TILE_F16_NHWC = 1
TILE_U8X4 = 2

# every symbol include/salve_hip.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = (
    "salve_hip_version",
    "salve_last_error",
    "salve_bev_workspace_bytes",
    "salve_bev_pano_index_bytes",
    "salve_bev_pano_index_build",
    "salve_bev_render_batch",
    "salve_bev_scatter",
    "salve_bev_densify",
    "salve_bev_scatter_points",
    "salve_zorder_winners",
    "salve_remove_hallucinated",
    "salve_bev_keys_from_pixels",
    "salve_bev_export_u8",
    "salve_layout_rasterise",
    "salve_bev_tiles",
    "salve_bev_tile_pairs",
    "salve_bev_densify_tiles",
    "salve_resize_rgb_u8",
    "salve_resnet_create",
    "salve_resnet_destroy",
    "salve_resnet_workspace_bytes",
    "salve_resnet_forward",
    "salve_resnet_num_layers",
)
# salve_resnet_create flags (include/salve_hip.h: SALVE_RESNET_*): kernel selection for the bit-identity tests; 0 = product
RESNET_CONV_IGEMM_ONLY, RESNET_CONV8_WHEREVER, RESNET_ROUND_ROBIN_TILES, RESNET_NO_STEM_FUSE, RESNET_NO_BLOCK_FUSE = 1, 2, 4, 8, 16
RESNET_NO_PROJ_FUSE, RESNET_NO_CHAIN, RESNET_CHAIN_EXPAND_ONLY, RESNET_CHAIN_16_WAVES, RESNET_CHAIN_NO_SPLIT = 32, 64, 128, 256, 512
RESNET_CHAIN_STORE_ALL, RESNET_NO_TRANSPOSED_TILES, RESNET_NO_NEXT_FUSE = 1024, 2048, 4096
STATUS_WALK_FAILED = 1
STATUS_FP16_RANGE = 2
STATUS_BAD_HYPOTHESIS = 4
STATUS_LAYOUT_THICKNESS = 8


class BevConfig(ctypes.Structure):
    """salve_bev_config_t"""

    _fields_ = [
        ("pano_h", ctypes.c_int32), ("pano_w", ctypes.c_int32), ("crop_rows", ctypes.c_int32),
        ("bev_h", ctypes.c_int32), ("bev_w", ctypes.c_int32), ("mask_k", ctypes.c_int32),
        ("depth_scale", ctypes.c_float), ("out_flags", ctypes.c_int32),
        ("win_xmin", ctypes.c_double), ("win_xmax", ctypes.c_double),
        ("win_ymin", ctypes.c_double), ("win_ymax", ctypes.c_double),
        ("img_tx", ctypes.c_double), ("img_ty", ctypes.c_double), ("img_scale", ctypes.c_double),
        ("rot_pre", ctypes.c_double * 4),
        ("z_lo", ctypes.c_double * 2), ("z_hi", ctypes.c_double * 2),
        ("z_min", ctypes.c_double), ("n_slices", ctypes.c_int32), ("reserved1", ctypes.c_int32),
    ]


HYP_DTYPE = np.dtype(
    [("pano_idx", "<i4"), ("surface", "<i4"), ("R", "<f4", (4,)), ("t", "<f4", (2,)), ("apply_pose", "<i4"),
     ("reserved", "<i4")]
)
TILE_JOB_DTYPE = np.dtype([("bev_offset", "<i8"), ("slot", "<i4"), ("chan", "<i4")])
LAYOUT_DTYPE = np.dtype([("n_poly", "<i4"), ("poly_off", "<i4"), ("n_seg", "<i4"), ("seg_off", "<i4")])
assert HYP_DTYPE.itemsize == 40 and TILE_JOB_DTYPE.itemsize == 16

_lib = None


class SalveHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the HIP library or raise -- never substitute a CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise SalveHipError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run __graft_entry__.build() "
            "(hipcc --offload-arch=gfx950). salve_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(str(LIB_PATH))
    vp, i32, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_size_t
    lib.salve_hip_version.restype = ctypes.c_int
    lib.salve_last_error.restype = ctypes.c_char_p
    lib.salve_bev_workspace_bytes.argtypes = [ctypes.POINTER(BevConfig), i32]
    lib.salve_bev_workspace_bytes.restype = sz
    lib.salve_bev_pano_index_bytes.argtypes = [ctypes.POINTER(BevConfig), i32]
    lib.salve_bev_pano_index_bytes.restype = sz
    lib.salve_bev_pano_index_build.argtypes = [ctypes.POINTER(BevConfig), vp, i32, vp, vp, sz, vp]
    lib.salve_bev_pano_index_build.restype = ctypes.c_int
    lib.salve_bev_render_batch.argtypes = [ctypes.POINTER(BevConfig), vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.salve_bev_render_batch.restype = ctypes.c_int
    lib.salve_bev_scatter.argtypes = [ctypes.POINTER(BevConfig), vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.salve_bev_scatter.restype = ctypes.c_int
    lib.salve_bev_densify.argtypes = [ctypes.POINTER(BevConfig), i32, vp, vp, vp, vp, vp, sz, vp]
    lib.salve_bev_densify.restype = ctypes.c_int
    lib.salve_bev_scatter_points.argtypes = [ctypes.POINTER(BevConfig), vp, vp, i32, vp, vp, vp, sz, vp]
    lib.salve_bev_scatter_points.restype = ctypes.c_int
    lib.salve_zorder_winners.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, vp, vp, vp]
    lib.salve_zorder_winners.restype = ctypes.c_int
    lib.salve_remove_hallucinated.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.salve_remove_hallucinated.restype = ctypes.c_int
    lib.salve_bev_keys_from_pixels.argtypes = [ctypes.POINTER(BevConfig), vp, vp, i32, vp, vp, sz, vp]
    lib.salve_bev_keys_from_pixels.restype = ctypes.c_int
    lib.salve_layout_rasterise.argtypes = [vp, i32, vp, vp, i32, i32, vp, vp, vp]
    lib.salve_layout_rasterise.restype = ctypes.c_int
    lib.salve_bev_export_u8.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.salve_bev_export_u8.restype = ctypes.c_int
    lib.salve_bev_tiles.argtypes = [vp, i32, i32, vp, i32, vp, vp, i32, i32, vp, vp, i32, i32, vp]
    lib.salve_bev_tiles.restype = ctypes.c_int
    lib.salve_bev_tile_pairs.argtypes = [vp, vp, i32, i32, vp, vp, i32, vp, vp, i32, i32, vp, vp, i32, i32, vp]
    lib.salve_bev_tile_pairs.restype = ctypes.c_int
    lib.salve_bev_densify_tiles.argtypes = [ctypes.POINTER(BevConfig), i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, vp, vp, sz, vp]
    lib.salve_bev_densify_tiles.restype = ctypes.c_int
    lib.salve_resize_rgb_u8.argtypes = [vp, i32, i32, i32, vp, i32, i32, vp, vp, vp]
    lib.salve_resize_rgb_u8.restype = ctypes.c_int
    lib.salve_resnet_create.argtypes = [i32, i32, vp, i32, vp, sz, vp, sz, vp, sz, i32]
    lib.salve_resnet_create.restype = vp
    lib.salve_resnet_destroy.argtypes = [vp]
    lib.salve_resnet_destroy.restype = None
    lib.salve_resnet_workspace_bytes.argtypes = [vp, i32]
    lib.salve_resnet_workspace_bytes.restype = sz
    lib.salve_resnet_forward.argtypes = [vp, vp, i32, vp, vp, sz, vp, vp]
    lib.salve_resnet_forward.restype = ctypes.c_int
    lib.salve_resnet_num_layers.argtypes = [vp]
    lib.salve_resnet_num_layers.restype = ctypes.c_int
    # The bindings above are written for ONE ABI: an older or newer library (a stale git-ignored .so, a SALVE_HIP_LIB override
    # built from another revision) would be called with shifted arguments -- device memory corruption instead of an error.
    got = int(lib.salve_hip_version())
    if got != EXPECTED_ABI:
        raise SalveHipError(f"{LIB_PATH} reports ABI version {got}, these bindings are written for {EXPECTED_ABI}: rebuild it "
                            "(python __graft_entry__.py --force)")
    _lib = lib
    return lib


def check_status_word(word: int, what: str) -> None:
    """Raise for a non-zero device status word (include/salve_hip.h: SALVE_STATUS_*)."""
    if word & STATUS_WALK_FAILED:
        raise SalveHipError(f"{what}: a Delaunay star walk did not close; the BEV image of at least one render is incomplete")
    if word & STATUS_FP16_RANGE:
        raise SalveHipError(f"{what}: an activation of the verifier exceeded the fp16 range and was saturated; the logits are "
                            "not those of the fp32 network (a network without trained normalisation statistics does this)")
    if word & STATUS_BAD_HYPOTHESIS:
        raise SalveHipError(f"{what}: a render row names a panorama outside the uploaded batch (or an unknown surface); its image is empty")
    if word & STATUS_LAYOUT_THICKNESS:
        raise SalveHipError(f"{what}: a layout segment of 19 pixels or more was left out (its OpenCV end caps are not implemented)")
    if word:
        raise SalveHipError(f"{what}: device status word {word:#x}")


def check(status: int, what: str) -> None:
    if status != SALVE_OK:
        msg = load().salve_last_error().decode("utf-8", "replace")
        raise SalveHipError(f"{what} failed with status {status}: {msg}")

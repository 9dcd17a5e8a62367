"""Host-side image I/O and panorama ingest (PIL; imageio / cv2 are not dependencies here).

Covers the file formats on either side of the hot path: RGB panorama JPEGs, uint16 `.depth.png` depth maps
(millimetres; producer salve/utils/infer_depth.py:55-62, consumer bev_rendering_utils.py:367) and the BEV JPEG tiles
(bev_rendering_utils.py:629-630).
"""

from __future__ import annotations

import os

import numpy as np
from PIL import Image


def read_rgb(path: str) -> np.ndarray:
    with Image.open(path) as im:
        if im.mode not in ("RGB", "L"):
            im = im.convert("RGB")
        return np.asarray(im).copy()


def read_depth_png(path: str) -> np.ndarray:
    with Image.open(path) as im:
        a = np.asarray(im)
    if a.dtype != np.uint16:
        a = a.astype(np.uint16)
    return a.copy()


def write_depth_png(path: str, depth_u16: np.ndarray) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(depth_u16.astype(np.uint16)).save(path)


def write_jpeg(path: str, img_u8: np.ndarray) -> None:
    """imageio.imwrite(path.jpg, img) == Pillow's JPEG encoder at its default quality (75)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(img_u8).save(path, quality=75)


def resize_linear_u8(img: np.ndarray, out_hw) -> np.ndarray:
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 images, as the reference applies to every
    panorama (bev_rendering_utils.py:375).  Identity when the size already matches; an exact 2x down-scale takes
    OpenCV's INTER_AREA fast path (2x2 box average, rounded: (a+b+c+d+2)>>2); otherwise the 11-bit fixed-point
    bilinear taps of salve_amd.rasteriser.linear_resize_taps."""
    h, w = out_hw
    H, W = img.shape[:2]
    if (H, W) == (h, w):
        return img
    if H == 2 * h and W == 2 * w:
        a = img.astype(np.uint16)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    from salve_amd.rasteriser import linear_resize_taps

    ty, tx = linear_resize_taps(h, H).astype(np.int64), linear_resize_taps(w, W).astype(np.int64)
    src = img.astype(np.int64)
    if src.ndim == 2:
        src = src[:, :, None]
    hor = src[:, tx[:, 0], :] * tx[None, :, 2, None] + src[:, tx[:, 1], :] * tx[None, :, 3, None]
    out = (((ty[:, 2, None, None] * (hor[ty[:, 0]] >> 4)) >> 16) + ((ty[:, 3, None, None] * (hor[ty[:, 1]] >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if img.ndim == 2 else out

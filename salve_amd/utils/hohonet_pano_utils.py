"""Equirectangular pixel -> unit-sphere direction table (HoHoNet frame).

Mirror of reference salve/utils/hohonet_pano_utils.py:10-44 (`get_uni_sphere_xyz`).
The table is separable -- x = cos(phi_v)*cos(theta_u), y = cos(phi_v)*sin(theta_u),
z = -sin(phi_v) -- so the device keeps only the four 1-D factors
(`get_sphere_factors`, 2H + 2W doubles = 24 KB at 512x1024) and forms the products
in the same order numpy does, which reproduces the reference's float64 table
bit for bit without ever calling device sin/cos.
"""

from __future__ import annotations

from typing import Tuple

import numpy as np


def get_sphere_factors(H: int, W: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Return (r[H], zdir[H], cos_t[W], sin_t[W]) as float64.

    The op order follows the reference exactly: theta = -(u + .5)/W, then *= 2*pi;
    phi = (v + .5)/H, then -= .5, then *= pi (hohonet_pano_utils.py:27-37).
    """
    u = np.arange(W)
    v = np.arange(H)
    theta = -(u + 0.5) / W
    theta *= 2 * np.pi
    phi = (v + 0.5) / H
    phi -= 0.5
    phi *= np.pi
    zdir = -np.sin(phi)
    r = np.cos(phi)
    return r, zdir, np.cos(theta), np.sin(theta)


def get_uni_sphere_xyz(H: int, W: int) -> np.ndarray:
    """[H, W, 3] float64 unit directions; -x points at the pano's centre pixel."""
    r, zdir, ct, st = get_sphere_factors(H, W)
    out = np.empty((H, W, 3), dtype=np.float64)
    out[..., 0] = r[:, None] * ct[None, :]
    out[..., 1] = r[:, None] * st[None, :]
    out[..., 2] = zdir[:, None]
    return out

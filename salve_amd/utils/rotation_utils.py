"""2-D rotation helpers on the hot path.  Mirror of salve/utils/rotation_utils.py:14-40 (no gtsam needed)."""

import numpy as np


def rotmat2d(theta_deg: float) -> np.ndarray:
    """[[c, -s], [s, c]] in float64.  cos(-90 deg) evaluates to 6.1e-17, not 0, and the rasteriser's
    bit-exact pixel indices depend on that value being carried through (bev_rendering_utils.py:443-446)."""
    theta = np.deg2rad(theta_deg)
    s, c = np.sin(theta), np.cos(theta)
    return np.array([[c, -s], [s, c]])


def rotmat2theta_deg(R: np.ndarray) -> float:
    return float(np.rad2deg(np.arctan2(R[1, 0], R[0, 0])))

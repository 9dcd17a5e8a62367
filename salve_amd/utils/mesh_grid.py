"""Regular grid as a point list.  Mirror of salve/utils/mesh_grid.py:11-36."""

import numpy as np


def get_mesh_grid_as_point_cloud(min_x: int, max_x: int, min_y: int, max_y: int, downsample_factor: float = 1.0) -> np.ndarray:
    """(N, 2) array of (x, y) grid coordinates, x varying fastest."""
    x = np.linspace(min_x, max_x, int((max_x - min_x + 1) / downsample_factor))
    y = np.linspace(min_y, max_y, int((max_y - min_y + 1) / downsample_factor))
    gx, gy = np.meshgrid(x, y)
    return np.hstack([gx.reshape(-1, 1), gy.reshape(-1, 1)])

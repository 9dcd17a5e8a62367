"""BEV grid parameters.  Mirror of the reference's salve/common/bevparams.py:17-99."""

from __future__ import annotations

import numpy as np

from salve_amd.common.sim2 import Sim2

DEFAULT_BEV_IMG_H_PX = 500
DEFAULT_BEV_IMG_W_PX = 500
DEFAULT_METERS_PER_PX = 0.02
FULL_RES_METERS_PER_PX = 0.005
FULL_RES_LINE_WIDTH_PX = 30


class BEVParams:
    """img_h x img_w pixels at meters_per_px; the rendered image is (img_h+1) x (img_w+1)
    (bev_rendering_utils.py:292-293) and covers [-lim, lim] metres with lim = int(half_px * m_per_px)."""

    def __init__(self, img_h: int = DEFAULT_BEV_IMG_H_PX, img_w: int = DEFAULT_BEV_IMG_W_PX,
                 meters_per_px: float = DEFAULT_METERS_PER_PX) -> None:
        self.img_h, self.img_w, self.meters_per_px = img_h, img_w, meters_per_px
        half_x = int((img_w / 2) * meters_per_px)
        half_y = int((img_h / 2) * meters_per_px)
        self.xlims = [-half_x, half_x]
        self.ylims = [-half_y, half_y]

    @property
    def bevimg_Sim2_world(self) -> Sim2:
        """p_img = (p_world - [xmin, ymin]) / meters_per_px."""
        return Sim2(R=np.eye(2), t=np.array([-self.xlims[0], -self.ylims[0]]), s=1 / self.meters_per_px)


def get_line_width_by_resolution(resolution: float) -> int:
    """Polyline width in px for a resolution (30 px at 0.005 m/px, never below 1)."""
    return max(round(FULL_RES_LINE_WIDTH_PX / (resolution / FULL_RES_METERS_PER_PX)), 1)

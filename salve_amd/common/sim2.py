"""Similarity(2) value type with the storage rules the rasteriser's bit-exact contract depends on.

Mirror of the reference's salve/common/sim2.py:23-199: rotation and translation are STORED AS FLOAT32
(sim2.py:50-51) and the scale as a Python float; `transform_from` computes s * (R p + t) (sim2.py:157-160).
Only the host-side API lives here -- the hot path receives R, t as float32 through salve_bev_hyp_t.
"""

from __future__ import annotations

import json
import os
from typing import Union

import numpy as np

PathLike = Union[str, "os.PathLike[str]"]


class Sim2:
    def __init__(self, R: np.ndarray, t: np.ndarray, s: Union[int, float]) -> None:
        for name, arr, shape in (("R", R, (2, 2)), ("t", t, (2,))):
            if not isinstance(arr, np.ndarray):
                raise ValueError(f"Input array `{name}` must be a Numpy n-d array.")
            if arr.shape != shape:
                raise ValueError(f"Input array `{name}` must have shape {shape}.")
        if not isinstance(s, (int, float)):
            raise AssertionError("scale must be an int or a float")
        if np.isclose(s, 0.0):
            raise ZeroDivisionError("3x3 matrix formation would require division by zero")
        self.R_ = R.astype(np.float32)
        self.t_ = t.astype(np.float32)
        self.s_ = float(s)

    # -- accessors
    @property
    def rotation(self) -> np.ndarray:
        return self.R_

    @property
    def translation(self) -> np.ndarray:
        return self.t_

    @property
    def scale(self) -> float:
        return self.s_

    @property
    def theta_deg(self) -> float:
        return float(np.rad2deg(np.arctan2(self.R_[1, 0], self.R_[0, 0])))

    @property
    def matrix(self) -> np.ndarray:
        T = np.zeros((3, 3))
        T[:2, :2], T[:2, 2], T[2, 2] = self.R_, self.t_, 1.0 / self.s_
        return T

    def __repr__(self) -> str:
        return f"Angle (deg.): {self.theta_deg:.1f}, Trans.: {np.round(self.t_, 2)}, Scale: {self.s_:.1f}"

    def __eq__(self, other: object) -> bool:
        return (
            isinstance(other, Sim2)
            and bool(np.isclose(self.s_, other.s_))
            and bool(np.allclose(self.R_, other.R_))
            and bool(np.allclose(self.t_, other.t_))
        )

    # -- group operations
    def compose(self, S: "Sim2") -> "Sim2":
        return Sim2(R=self.R_ @ S.R_, t=self.R_ @ S.t_ + ((1.0 / S.s_) * self.t_), s=self.s_ * S.s_)

    def inverse(self) -> "Sim2":
        Rt = self.R_.T
        return Sim2(Rt, -Rt @ (self.s_ * self.t_), 1.0 / self.s_)

    def transform_from(self, point_cloud: np.ndarray) -> np.ndarray:
        if not isinstance(point_cloud, np.ndarray):
            raise ValueError("Input `point_cloud` must be a Numpy n-d array.")
        if point_cloud.ndim != 2:
            raise ValueError("Input point cloud is not 2-dimensional.")
        if point_cloud.shape[1] != 2:
            raise ValueError("Input `point_cloud` must have shape (N,2).")
        return ((point_cloud @ self.R_.T) + self.t_) * self.s_

    transform_point_cloud = transform_from

    # -- I/O: {"R": [4], "t": [2], "s": float}  (sim2.py:165-188)
    def save_as_json(self, save_fpath: PathLike) -> None:
        os.makedirs(os.path.dirname(os.path.abspath(save_fpath)), exist_ok=True)
        with open(save_fpath, "w") as f:
            json.dump({"R": self.R_.flatten().tolist(), "t": self.t_.flatten().tolist(), "s": self.s_}, f)

    @classmethod
    def from_json(cls, json_fpath: PathLike) -> "Sim2":
        with open(json_fpath, "r") as f:
            d = json.load(f)
        return cls(np.array(d["R"]).reshape(2, 2), np.array(d["t"]).reshape(2), float(d["s"]))

    @classmethod
    def from_matrix(cls, T: np.ndarray) -> "Sim2":
        if np.isclose(T[2, 2], 0.0):
            raise ZeroDivisionError("Sim(2) scale calculation would lead to division by zero.")
        return cls(T[:2, :2], T[:2, 2], 1 / T[2, 2])

"""Validation / test image transform on the GPU: Resize (cv2 INTER_LINEAR, uint8) -> centre Crop -> ToTensor ->
Normalize, i.e. what salve/train_utils.py:126-159 composes from salve/utils/transform.py:256-272, 386-420, 105-123,
177-202 -- one launch of the tile kernel (salve_bev_tiles) per call instead of four numpy / cv2 passes.

The callable takes the 2 / 4 / 6 HWC uint8 images of one example (as the reference's Pair / Quadruplet / Sextuplet
transforms do) and returns as many float32 [3, crop_h, crop_w] tensors, on the GPU.  No CPU fallback.
"""

from __future__ import annotations

import ctypes
from typing import Dict, Tuple

import numpy as np
import torch

from salve_amd import _lib
from salve_amd.rasteriser import linear_resize_taps, normalisation_lut


class ValTestTransform:
    def __init__(self, resize_hw: Tuple[int, int], crop_hw: Tuple[int, int], device=None) -> None:
        if resize_hw[0] != resize_hw[1] or crop_hw[0] != crop_hw[1]:
            raise RuntimeError("the tile kernel resizes / crops to squares (resize_h == resize_w, train_h == train_w)")
        if crop_hw[0] > resize_hw[0]:
            raise RuntimeError("centre crop larger than the resized image (the reference would pad with the mean)")
        self.resize, self.crop = int(resize_hw[0]), int(crop_hw[0])
        self.device = device
        self._taps: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor]] = {}
        self._lut = None

    def _dev(self) -> torch.device:
        if self.device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("salve_amd.transforms needs the MI355X (no CPU fallback); pass transform=None to read raw tiles")
            self.device = torch.device("cuda", torch.cuda.current_device())
        return torch.device(self.device)

    def __call__(self, *images: np.ndarray):
        dev = self._dev()
        lib = _lib.load()
        h, w = images[0].shape[:2]
        for im in images:
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3 or im.shape[:2] != (h, w):
                raise RuntimeError("expected equally sized HWC uint8 RGB images")
        if (h, w) not in self._taps:
            self._taps[(h, w)] = (torch.from_numpy(linear_resize_taps(self.resize, h)).to(dev),
                                  torch.from_numpy(linear_resize_taps(self.resize, w)).to(dev))
        if self._lut is None:
            self._lut = torch.from_numpy(normalisation_lut()).to(dev)
        coef_y, coef_x = self._taps[(h, w)]
        k = len(images)
        stack = np.stack(images).astype(np.uint32)
        packed = torch.from_numpy((stack[..., 0] | (stack[..., 1] << 8) | (stack[..., 2] << 16)).astype(np.int32)).to(dev)  # 0x00BBGGRR
        jobs = np.zeros(k, dtype=_lib.TILE_JOB_DTYPE)
        jobs["bev_offset"] = np.arange(k, dtype=np.int64) * (h * w)
        jobs["slot"] = np.arange(k, dtype=np.int32)
        jobs["chan"] = 0
        jobs_dev = torch.from_numpy(jobs.view(np.uint8)).to(dev)
        out = torch.empty((k, 3, self.crop, self.crop), dtype=torch.float32, device=dev)
        st = lib.salve_bev_tiles(ctypes.c_void_p(packed.data_ptr()), h, w, ctypes.c_void_p(jobs_dev.data_ptr()), k,
                                 ctypes.c_void_p(coef_y.data_ptr()), ctypes.c_void_p(coef_x.data_ptr()), self.resize, self.crop,
                                 ctypes.c_void_p(self._lut.data_ptr()), ctypes.c_void_p(out.data_ptr()), _lib.TILE_F32_NCHW, 3,
                                 ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(st, "salve_bev_tiles")
        return tuple(out[i] for i in range(k))

"""Seeded synthetic panoramas, depth maps and Sim(2) hypotheses.

There is no dataset on the GPU box, so tests, golden vectors and ``bench.py`` all
draw their inputs from here.  Everything is plain numpy and deterministic for a
given (pano index, H, W) / (seed, N, P).

* depth: ray-cast of an axis-aligned box room seen from the origin, in the
  HoHoNet pano frame (-x towards the pano centre column, see
  ``hohonet_pano_utils.get_uni_sphere_xyz``), quantised to uint16 millimetres
  -- the on-disk ``.depth.png`` format of the reference
  (reference: salve/utils/infer_depth.py:55-62, read back at
  salve/utils/bev_rendering_utils.py:367).
* rgb: uint8 [H, W, 3]; smooth sinusoids + uniform noise.  Even pano indices are
  clamped to >= 1 in every channel, odd ones are unrestricted so that colours
  whose uint8 channel product wraps to zero occur (the mask rule of
  salve/utils/interpolation_utils.py:95-98).
* hypotheses: theta ~ U[0, 360) deg, t ~ U[-2, 2]^2, s = 1, stored as float32
  like ``Sim2`` does (reference: salve/common/sim2.py:50-52).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from salve_amd.utils.hohonet_pano_utils import get_uni_sphere_xyz

FLOOR_Z = -1.5
CEILING_Z = 1.2
WALL_X = (-2.5, 3.0)
WALL_Y = (-2.0, 3.5)
WALL_JITTER = 0.5


def make_box_room_depth_mm(pano_idx: int, H: int = 512, W: int = 1024) -> np.ndarray:
    """uint16 [H, W] depth in millimetres of a jittered box room."""
    rng = np.random.default_rng(seed=pano_idx)
    jit = rng.uniform(-WALL_JITTER, WALL_JITTER, size=4)
    x0, x1 = WALL_X[0] + jit[0], WALL_X[1] + jit[1]
    y0, y1 = WALL_Y[0] + jit[2], WALL_Y[1] + jit[3]
    d = get_uni_sphere_xyz(H, W)  # unit directions [H, W, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(d[..., 0] > 0, x1 / d[..., 0], np.where(d[..., 0] < 0, x0 / d[..., 0], np.inf))
        ty = np.where(d[..., 1] > 0, y1 / d[..., 1], np.where(d[..., 1] < 0, y0 / d[..., 1], np.inf))
        tz = np.where(d[..., 2] > 0, CEILING_Z / d[..., 2], np.where(d[..., 2] < 0, FLOOR_Z / d[..., 2], np.inf))
    t = np.minimum(np.minimum(tx, ty), tz)
    mm = np.round(t * 1000.0)
    return np.clip(mm, 0, 65535).astype(np.uint16)


def make_pano_rgb(pano_idx: int, H: int = 512, W: int = 1024) -> np.ndarray:
    """uint8 [H, W, 3] synthetic panorama texture."""
    rng = np.random.default_rng(seed=1_000_003 + pano_idx)
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    ph = rng.uniform(0, 2 * np.pi, size=6)
    fr = rng.uniform(3.0, 11.0, size=6)
    chans = []
    for c in range(3):
        base = 127.5 + 90.0 * np.sin(2 * np.pi * fr[2 * c] * u / W + ph[2 * c]) * np.cos(
            2 * np.pi * fr[2 * c + 1] * v / H + ph[2 * c + 1]
        )
        chans.append(base)
    img = np.stack(chans, axis=-1) + rng.uniform(-37.0, 37.0, size=(H, W, 3))
    img = np.clip(np.round(img), 0, 255).astype(np.uint8)
    if pano_idx % 2 == 0:
        img = np.maximum(img, 1)
    return img


def make_cluttered_room_depth_mm(pano_idx: int, H: int = 512, W: int = 1024) -> np.ndarray:
    """uint16 [H, W] depth in millimetres of the jittered box room of `make_box_room_depth_mm` with 3-5 occluding boxes
    standing on the floor (furniture: 0.4-1.1 m wide, 0.4-1.9 m tall, never over the camera) and a door opening in the +x
    wall (0.9 m wide, 2.0 m tall) through which the rays continue into a 2 m deep corridor.  The floor and ceiling clouds of
    this scene have shadows behind the furniture and a notch at the door: a NON-convex outline with interior holes, which is
    what raises the share of sites the lean star walk hands to the general walk."""
    rng = np.random.default_rng(seed=pano_idx)
    jit = rng.uniform(-WALL_JITTER, WALL_JITTER, size=4)          # same walls as the box room of this pano index
    x0, x1 = WALL_X[0] + jit[0], WALL_X[1] + jit[1]
    y0, y1 = WALL_Y[0] + jit[2], WALL_Y[1] + jit[3]
    rng = np.random.default_rng(seed=7_000_003 + pano_idx)
    d = get_uni_sphere_xyz(H, W)
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = [np.where(c != 0, 1.0 / c, np.inf) for c in (dx, dy, dz)]
        # room shell; the +x wall has a door opening y in [ya, ya + 0.9], z below FLOOR_Z + 2.0, behind it a corridor
        ya = rng.uniform(y0 + 0.3, y1 - 1.2)
        corridor = 2.0
        tz = np.where(dz > 0, CEILING_Z * inv[2], np.where(dz < 0, FLOOR_Z * inv[2], np.inf))
        ty = np.where(dy > 0, y1 * inv[1], np.where(dy < 0, y0 * inv[1], np.inf))
        tx_near = np.where(dx > 0, x1 * inv[0], np.where(dx < 0, x0 * inv[0], np.inf))
        hit_y, hit_z = tx_near * dy, tx_near * dz
        through = (dx > 0) & (hit_y >= ya) & (hit_y <= ya + 0.9) & (hit_z <= FLOOR_Z + 2.0)
        # inside the corridor the side walls are the door jambs, the far wall is at x1 + corridor
        ty_corr = np.where(dy > 0, (ya + 0.9) * inv[1], np.where(dy < 0, ya * inv[1], np.inf))
        tz_corr = np.where(dz > 0, (FLOOR_Z + 2.0) * inv[2], np.where(dz < 0, FLOOR_Z * inv[2], np.inf))
        t_corr = np.minimum(np.minimum((x1 + corridor) * inv[0], ty_corr), tz_corr)
        t = np.minimum(np.minimum(np.where(through, np.inf, tx_near), ty), tz)
        t = np.where(through & (tx_near <= np.minimum(ty, tz)), t_corr, t)
        # furniture: axis-aligned boxes on the floor, slab test (the camera is outside every box)
        for _ in range(int(rng.integers(3, 6))):
            w, dpt, h = rng.uniform(0.4, 1.1), rng.uniform(0.4, 1.1), rng.uniform(0.4, 1.9)
            for _try in range(20):
                bx, by = rng.uniform(x0 + 0.1, x1 - w - 0.1), rng.uniform(y0 + 0.1, y1 - dpt - 0.1)
                if not (bx - 0.35 < 0 < bx + w + 0.35 and by - 0.35 < 0 < by + dpt + 0.35):
                    break
            else:
                continue
            lo = np.array([bx, by, FLOOR_Z])
            hi = np.array([bx + w, by + dpt, min(FLOOR_Z + h, CEILING_Z - 0.05)])
            t_in, t_out = np.full(dx.shape, -np.inf), np.full(dx.shape, np.inf)
            for a in range(3):
                ta, tb = lo[a] * inv[a], hi[a] * inv[a]
                par = ~np.isfinite(inv[a])                      # ray parallel to this slab: origin 0 must lie inside it
                inside = (lo[a] <= 0.0) & (0.0 <= hi[a])
                t_in = np.where(par, np.where(inside, t_in, np.inf), np.maximum(t_in, np.minimum(ta, tb)))
                t_out = np.where(par, t_out, np.minimum(t_out, np.maximum(ta, tb)))
            hit = (t_in < t_out) & (t_in > 0)
            t = np.where(hit, np.minimum(t, t_in), t)
    mm = np.round(t * 1000.0)
    return np.clip(mm, 0, 65535).astype(np.uint16)


def make_noisy_room_depth_mm(pano_idx: int, H: int = 512, W: int = 1024) -> np.ndarray:
    """uint16 [H, W]: the cluttered scene seen through a depth NETWORK instead of a ray caster -- what the reference's renderer is
    actually fed (HoHoNet predictions, salve/utils/infer_depth.py:55-62).  Two error terms on the exact depth: a smooth
    multiplicative bias field (+-2.5 %: a few low-frequency sinusoids over the panorama -- planes that bend and tilt) and per-pixel
    noise (sigma 0.6 %, i.e. ~1.5 cm at 2.5 m), plus a small fraction of outliers at depth discontinuities left to the z filter.
    The floor / ceiling clouds of this scene are ragged: neighbouring panorama pixels no longer land on neighbouring BEV pixels in
    order, rows interleave, holes and isolated sites appear inside the cloud -- more outline, more hard sites for the general star
    walk than the box room or the cluttered scene (the headline is measured on the easiest input)."""
    exact = make_cluttered_room_depth_mm(pano_idx, H, W).astype(np.float64)
    rng = np.random.default_rng(seed=11_000_003 + pano_idx)
    v, u = np.meshgrid(np.arange(H, dtype=np.float64) / H, np.arange(W, dtype=np.float64) / W, indexing="ij")
    bias = np.zeros((H, W))
    for _ in range(4):
        fu, fv = int(rng.integers(1, 5)), int(rng.integers(1, 4))
        bias += rng.uniform(0.004, 0.009) * np.sin(2 * np.pi * (fu * u + rng.uniform()) ) * np.cos(2 * np.pi * (fv * v + rng.uniform()))
    noisy = exact * (1.0 + bias + rng.normal(0.0, 0.006, size=(H, W)))
    # a network smears depth discontinuities: 1 % of the pixels take the mean of their own and their right neighbour's depth
    smear = rng.random((H, W)) < 0.01
    noisy = np.where(smear, 0.5 * (noisy + np.roll(noisy, -1, axis=1)), noisy)
    return np.clip(np.round(noisy), 0, 65535).astype(np.uint16)


SCENES = {"box": make_box_room_depth_mm, "cluttered": make_cluttered_room_depth_mm, "noisy": make_noisy_room_depth_mm}


def make_pano(pano_idx: int, H: int = 512, W: int = 1024, scene: str = "box") -> Tuple[np.ndarray, np.ndarray]:
    """(rgb uint8 [H,W,3], depth uint16 [H,W]).  scene: "box" (the benchmark's scene), "cluttered" (occluders + a door)
    or "noisy" (the cluttered scene with network-like depth errors)."""
    return make_pano_rgb(pano_idx, H, W), SCENES[scene](pano_idx, H, W)


@dataclass
class HypothesisTable:
    """N alignment hypotheses i2Ti1 over P panoramas (structure of arrays)."""

    i1: np.ndarray  # int32 [N] pano rendered under the pose
    i2: np.ndarray  # int32 [N] pano rendered at identity
    R: np.ndarray  # float32 [N, 2, 2]
    t: np.ndarray  # float32 [N, 2]
    theta_deg: np.ndarray  # float64 [N]
    # bool [N] or None: True where the verifier must see (pano i2, pano i1) instead of (i1, i2).  The reference's dataset
    # orders the two tiles of a pair by FILE NAME (salve/dataset/zind_data.py:110 `pair_fpaths.sort()`), i.e. by pano stem,
    # not by (i1, i2); released checkpoints were trained that way.  Synthetic tables have no names: no swap.
    swap: Optional[np.ndarray] = None

    def __len__(self) -> int:
        return int(self.i1.shape[0])

    @staticmethod
    def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
        """Contiguous block split, rank r gets rows [r*N/G, (r+1)*N/G); sizes differ by at most one."""
        return (rank * n) // world, ((rank + 1) * n) // world

    def shard(self, rank: int, world: int) -> "HypothesisTable":
        lo, hi = self.shard_bounds(len(self), rank, world)
        return HypothesisTable(self.i1[lo:hi], self.i2[lo:hi], self.R[lo:hi], self.t[lo:hi], self.theta_deg[lo:hi],
                               None if self.swap is None else self.swap[lo:hi])


def make_hypotheses(n: int, num_panos: int, seed: int = 0) -> HypothesisTable:
    rng = np.random.default_rng(seed)
    theta = rng.uniform(0.0, 360.0, size=n)
    t = rng.uniform(-2.0, 2.0, size=(n, 2))
    i1 = rng.integers(0, num_panos, size=n)
    if num_panos > 1:
        off = rng.integers(1, num_panos, size=n)
        i2 = (i1 + off) % num_panos
    else:
        i2 = i1.copy()
    th = np.deg2rad(theta)
    c, s = np.cos(th), np.sin(th)
    R = np.stack([np.stack([c, -s], -1), np.stack([s, c], -1)], -2)
    return HypothesisTable(
        i1.astype(np.int32), i2.astype(np.int32), R.astype(np.float32), t.astype(np.float32), theta
    )


def make_panos(n: int, H: int = 512, W: int = 1024, scene: str = "box", threads: int = 8):
    """[(rgb, depth)] of panoramas 0 .. n-1, generated by a few host threads (numpy releases the GIL; a 2048 x 1024 panorama takes 2 s)."""
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=max(1, min(threads, n))) as pool:
        return list(pool.map(lambda i: make_pano(i, H, W, scene=scene), range(n)))


def trained_looking_batchnorm(model, seed: int = 0) -> None:
    """Give a freshly initialised verifier BatchNorm statistics that look like a trained network's, so that activations stay
    O(1) through the trunk.  (torchvision's initialisation -- weight 1, bias 0, mean 0, variance 1 -- is no normalisation
    at all: activations grow with depth, up to 1e8 in ResNet-152.  No checkpoint is available offline, so tests, smoke() and
    the benchmark use these seeded statistics; the last BatchNorm of every block is scaled down the way training leaves it.)"""
    import torch

    g = torch.Generator().manual_seed(seed)
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            last = name.endswith("bn3") or (name.endswith("bn2") and model.resnet.block_kind == "basic")
            m.weight.data = (0.25 if last else 1.0) * (0.6 + 0.4 * torch.rand(m.num_features, generator=g))
            m.bias.data = 0.1 * torch.randn(m.num_features, generator=g)
            m.running_mean.data = 0.1 * torch.randn(m.num_features, generator=g)
            m.running_var.data = 0.6 + 0.8 * torch.rand(m.num_features, generator=g)


def trained_looking_head(model, scale: float = 30.0) -> None:
    """Scale the classifier head so that the logits have the magnitude a trained SALVe verifier emits (|logit| of several units:
    its softmax outputs are mostly above 0.9, scripts/test.py:217-229) while the trunk's activations stay O(1): torch's default
    `nn.Linear` initialisation (uniform +-1/sqrt(2048)) gives |logit| 0.2-0.4 on `trained_looking_batchnorm` networks, and an
    ABSOLUTE logit bound checked only there says nothing about fp16 storage at |logit| 5-10.  scale 30 ->
    |logit| up to ~5 (ResNet-50) / ~11 (ResNet-152, 12 channels) on tile-like input."""
    model.fc.weight.data = model.fc.weight.data * float(scale)

"""Development: layout kernel against the oracle, with the differing pixels listed (GPU box)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from oracle import layout_oracle as lo
from salve_amd import layout
from salve_amd.rasteriser import BevRasteriser
rng = np.random.default_rng(3)
specs = []
for k in range(6):
    n = int(rng.integers(4, 12))
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rad = rng.uniform(0.8, 3.2, n)
    room = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1) + rng.uniform(-1, 1, 2)
    wdos = []
    for j in range(int(rng.integers(0, 6))):
        a = int(rng.integers(0, n))
        p, q = room[a], room[(a + 1) % n]
        t0, t1 = np.sort(rng.uniform(0, 1, 2))
        wdos.append((("doors", "windows", "openings")[j % 3], np.stack([p + t0 * (q - p), p + t1 * (q - p)])))
    specs.append((np.vstack([room, room[:1]]), wdos))
specs.append((np.array([[-9.0, -9.0], [9.0, -9.0], [9.0, 9.0], [-9.0, 9.0]]), [("doors", np.array([[-20.0, 0.0], [20.0, 0.3]]))]))
dev = torch.device("cuda:0")
u8 = BevRasteriser(dev).export_u8(layout.rasterise_layouts(specs, dev)).cpu().numpy()
for k, (room, wdos) in enumerate(specs):
    exp = lo.rasterize_single_layout(room, wdos)
    d = (u8[k] != exp).any(-1)
    print(f"layout {k}: {len(wdos)} segments, {int(d.sum())} pixels differ")
    if d.any():
        for wt, v in wdos:
            print("   seg", wt, (lo.to_pixels(v * 1.5)).tolist())
        ys, xs = np.nonzero(d)
        for y, x in list(zip(ys, xs))[:24]:
            print(f"   (x {x}, y_img {500 - y}): gpu {u8[k][y, x].tolist()} oracle {exp[y, x].tolist()}")

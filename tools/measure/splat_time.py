"""Development: time the scatter stage (panorama index + splat) and the densify stage at the benchmark's launch shape.
    python tools/measure/splat_time.py [n] [panos] [sorted]      (SALVE_HIP_LIB selects an ablation build)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
srt = len(sys.argv) > 3 and sys.argv[3] == "sorted"
dev = torch.device("cuda:0")
ras = BevRasteriser(dev)
panos = [synthetic.make_pano(i) for i in range(P)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
hyp = synthetic.make_hypotheses(n, P, seed=0)
order = np.argsort(hyp.i1, kind="stable") if srt else np.arange(n)
hd = ras.upload_hypotheses(pack_hypotheses(hyp.i1[order], np.zeros(n), hyp.R[order], hyp.t[order], np.ones(n)))
bev = torch.empty((n,) + ras.bev_hw, dtype=torch.int32, device=dev)

def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def build_index():
    ras.drop_pano_index(d_depth); ras.pano_index(d_depth)

t_idx = timed(build_index)
t_sc = timed(lambda: ras.scatter(d_rgb, d_depth, hd, n, bev))
import os
t_de = float("nan") if os.environ.get("SALVE_HIP_LIB") else timed(lambda: ras.densify(n, bev))   # (an ablated scatter leaves no valid bitmaps)
cnt = torch.zeros(n, dtype=torch.int32, device=dev)
ras.scatter(d_rgb, d_depth, hd, n, bev, in_window=cnt)
print("in_window counter: mean", float(cnt.float().mean()), "(ablation build 1: blocks that reach a tile, summed over the render's tiles)")
print(f"n={n} panos={P} sorted={srt}: index build {t_idx:.3f} ms ({P} panoramas), scatter {t_sc:.3f} ms, densify {t_de:.3f} ms")

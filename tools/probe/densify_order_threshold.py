"""Development (GPU box): the densify stage with the costly renders first (default) and in the given order (out_flags bit 4), for launches
of 640 ... 4096 renders -- where the ordering starts to pay (bev_render.hip: ORDER_MIN_RENDERS).   usage: densify_order_threshold.py [scene]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses
scene = sys.argv[1] if len(sys.argv) > 1 else "box"
dev = torch.device("cuda:0")
P = 64
ras = BevRasteriser(dev)
panos = [synthetic.make_pano(i, scene=scene) for i in range(P)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
hyp = synthetic.make_hypotheses(4096, P, seed=0)
order = np.argsort(hyp.i1, kind="stable")
for n in (640, 768, 1024, 1536, 2048, 3072, 4096):
    sel = np.sort(order[:n]) if False else order[np.linspace(0, 4095, n).astype(int)]
    hd = ras.upload_hypotheses(pack_hypotheses(hyp.i1[sel], np.zeros(n), hyp.R[sel], hyp.t[sel], np.ones(n)))
    buf = torch.empty((n, *ras.bev_hw), dtype=torch.int32, device=dev)
    res = {}
    for flag in (0, 4, 0, 4):
        ras.cfg.out_flags = flag
        ts = []
        for rep in range(5):
            ras.scatter(d_rgb, d_depth, hd, n, buf)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ras.densify(n, buf); e1.record()
            torch.cuda.synchronize()
            if rep: ts.append(e0.elapsed_time(e1))
        res.setdefault(flag, []).append(np.mean(ts))
    ras.cfg.out_flags = 0
    print(f"{scene} n={n:5d}: costly first {res[0][0]:.3f} {res[0][1]:.3f} ms   as given {res[4][0]:.3f} {res[4][1]:.3f} ms   ({(np.mean(res[0]) / np.mean(res[4]) - 1) * 100:+.1f} %)", flush=True)

"""Stand-alone render of a few BEV images with debug buffers; prints per-render stats (GPU box only)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from salve_amd import synthetic
from salve_amd.rasteriser import BevRasteriser, pack_hypotheses

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
ras = BevRasteriser(dev)
import os
ras.cfg.reserved1 = int(os.environ.get("SALVE_DBG_FLAGS", "0"))
panos = [synthetic.make_pano(i) for i in range(2)]
d_rgb, d_depth = ras.upload_panos(np.stack([p[0] for p in panos]), np.stack([p[1] for p in panos]))
hyp = synthetic.make_hypotheses(max(n, 16), 2, seed=0)
h = pack_hypotheses(hyp.i1[:n], np.arange(n) % 2, hyp.R[:n], hyp.t[:n], np.ones(n))
hd = ras.upload_hypotheses(h)
print("launch", flush=True)
bev, dbg = ras.render(d_rgb, d_depth, hd, n, debug=True)
torch.cuda.synchronize()
print("stats", dbg.stats.cpu().numpy()[:4], flush=True)
t0 = time.time()
for _ in range(3):
    bev, _ = ras.render(d_rgb, d_depth, hd, n)
torch.cuda.synchronize()
print("ms per render", (time.time() - t0) / 3 / n * 1e3)
